"""The assembly K loop (csrc/conv_kloop_gfx950.inc) is generated: the committed file must be what tools/gen_conv_kloop.py prints, and the
schedule it encodes must keep the properties the kernel relies on (no GPU needed: text checks)."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "coupe", "optical_flow_based_deep_video_stabilization_amd", "csrc", "conv_kloop_gfx950.inc")


def _gen():
    spec = importlib.util.spec_from_file_location("gen_conv_kloop", os.path.join(ROOT, "tools", "gen_conv_kloop.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_committed_include_is_the_generators_output():
    assert open(INC).read() == _gen().render()


def _blocks():
    """tile shape (BM, BN) -> the K loop's lines: 128 x 128, 128 x 64 (2 x 2 waves) and 64 x 128 (1 x 4 waves, the one-sample launches)"""
    g = _gen()
    out = {(128, bn): g.Gen(bn).generate() for bn in (128, 64)}
    out[(64, 128)] = g.Gen(128, BM=64, WM=1, WN=4).generate()
    return out


def test_every_tile_body_has_the_full_mfma_count_and_one_barrier():
    for (bm, bn), lines in _blocks().items():
        nb = bn // 64 if bm == 128 else 1
        per_tile = 4 * 4 * 2 * nb                                           # 4 k-groups x 4 j x MB x NB
        text = "\n".join(lines)
        bodies = re.split(r"\.Lvk\d+_(?:body0|body1|tail0|tail1|end)_%=:", text)[1:5]
        assert len(bodies) == 4
        for i, b in enumerate(bodies):
            assert b.count("v_mfma_f32_32x32x2_f32") == per_tile
            assert b.count("s_barrier") == (1 if i < 2 else 0)                  # steady-state bodies only; the last tile fetches nothing
            assert b.count("buffer_load_dwordx4") == ((bm // 32 + bn // 32) if i < 2 else 0)
    # the three shapes' labels are distinct (one translation unit instantiates all of them)
    tags = {re.search(r"\.Lvk(\d+)_body0", "\n".join(l)).group(1) for l in _blocks().values()}
    assert tags == {"128", "64", "64128"}


def test_exec_is_whole_again_before_every_lds_dma_load_and_no_mfma_runs_under_a_narrowed_exec():
    for lines in _blocks().values():
        narrowed = False
        for l in lines:
            if l.startswith("v_cmpx"):
                narrowed = True
            elif l.startswith("s_mov_b64 exec, -1"):
                narrowed = False
            elif l.startswith(("buffer_load", "v_mfma", "ds_read", "s_barrier", "s_cbranch", "s_branch")):
                assert not narrowed, l


def test_m0_is_written_at_least_one_instruction_before_the_load_that_uses_it():
    for lines in _blocks().values():
        for i, l in enumerate(lines):
            if l.startswith("buffer_load"):
                assert not lines[i - 1].startswith("s_add_i32 m0"), (lines[i - 1], l)          # 1 wait state: SALU M0 write -> LDS-DMA
                assert any(x.startswith("s_add_i32 m0") for x in lines[max(0, i - 10):i])


def test_accumulator_chain_order_is_k_group_then_j():
    """per accumulator the A/B fragment registers come in the order the C++ loop multiplies them (bit-identical results)"""
    g = _gen()
    for gen in (g.Gen(128), g.Gen(64), g.Gen(128, BM=64, WM=1, WN=4)):
        for s in (0, 1):
            seq = {}
            for l in gen.mfmas(s):
                m = re.match(r"v_mfma_f32_32x32x2_f32 %\[(c\d\d)\], v(\d+), v(\d+),", l)
                seq.setdefault(m.group(1), []).append((int(m.group(2)), int(m.group(3))))
            assert len(seq) == gen.MB * gen.NB
            for acc, ops in seq.items():
                mb, nb = int(acc[1]), int(acc[2])
                a0, b0 = g.frag(s, f"A{mb}", 0), g.frag(s, f"B{nb}", 0)
                assert ops == [(a0 + j, b0 + j) for j in range(4)]


def test_wgrad_blocks_keep_the_same_invariants():
    g = _gen()
    for bn in (128, 64):
        gen = g.WgradGen(bn)
        lines = gen.generate()
        text = "\n".join(lines)
        bodies = re.split(r"\.Lvwg\d+_(?:body0|body1|tail0|tail1|end)_%=:", text)[1:5]
        per_tile = 16 * 2 * (bn // 64)                                       # 16 k-steps x MB x NB
        for i, b in enumerate(bodies):
            assert b.count("v_mfma_f32_32x32x2_f32") == per_tile
            assert b.count("s_barrier") == (1 if i < 2 else 0)
            assert b.count("buffer_load_dwordx4") == ((4 + gen.BP) if i < 2 else 0)
            assert b.count("global_load_dwordx3") == (4 if i < 2 else 0)       # the pixel-table entries of the tile after the next one
        narrowed = False
        for l in lines:
            if l.startswith("v_cmpx") or (l.startswith("s_mov_b64 exec,") and "-1" not in l):
                narrowed = True
            elif l.startswith("s_mov_b64 exec, -1"):
                narrowed = False
            elif l.startswith(("buffer_load", "global_load", "v_mfma", "ds_read", "s_barrier", "s_cbranch", "s_branch")):
                assert not narrowed, l
        # per accumulator the fragments come in k-step order
        for s in (0, 1):
            seq = {}
            for l in gen.mfmas(s):
                m = re.match(r"v_mfma_f32_32x32x2_f32 %\[(c\d\d)\], v(\d+), v(\d+),", l)
                seq.setdefault(m.group(1), []).append((int(m.group(2)), int(m.group(3))))
            base = gen.FR + 16 * s
            for acc, ops in seq.items():
                mb, nb = int(acc[1]), int(acc[2])
                assert ops == [(base + 4 * mb + i, base + 8 + 4 * nb + i) for i in range(4)]


def test_stream_blocks_keep_their_invariants():
    """stream forms (first layer, Winograd-domain GEMMs): every tile's MFMA count, no memory instruction or MFMA under a narrowed
    EXEC, all 32 accumulators flushed at every seam, every flushed value stored exactly once per seam"""
    g = _gen()
    rs = g.RowWinStreamGen(6).generate()
    text = "\\n".join(rs)
    assert text.count("v_accvgpr_read_b32") == 2 * 32                          # the seam's flush and the final one
    assert text.count("buffer_store_dword") == 32
    assert text.count("v_mfma_f32_32x32x2_f32") == 5 * 6 * 32                   # five row bodies (first, more, bridge, first + stores, final)
    assert sum(1 for l in rs if l.startswith("v_mfma") and l.endswith(", 0")) == 2 * 2      # two accumulators x two zero-start rows
    gs = g.GemmStreamGen().generate()
    text = "\\n".join(gs)
    assert text.count("v_accvgpr_read_b32") == 2 * 32
    assert text.count("buffer_store_dword") == 32
    assert text.count("v_mfma_f32_32x32x2_f32") == 6 * 32                       # first, first + stores, mid x 2, last, final
    assert text.count("s_barrier") == 5                                         # every body but the final one
    for lines in (rs, gs):
        for i, l in enumerate(lines):
            if l.startswith("buffer_load") and l.endswith("lds"):
                assert any(x.startswith("s_add_i32 m0") for x in lines[max(0, i - 4):i]) and not lines[i - 1].startswith("s_add_i32 m0")
