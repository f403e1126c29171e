#!/usr/bin/env python3
"""Writes csrc/conv_kloop_gfx950.inc: the steady-state K loop of conv_mfma_kernel<128, BN, 2, 2, true, true> (BN = 128, 64) as one
gfx950 assembly block per tile shape -- explicit issue order, s_waitcnt placement and barrier position.

    python tools/gen_conv_kloop.py            # rewrites the .inc (tests/test_kloop_gen.py checks the committed file is what this prints)

What one K-tile (32 floats of K, tile t in LDS buffer b, tile t+1 fetched into buffer b^1) looks like for a wave
(MB x NB blocks of 32x32, G = 4*MB*NB MFMAs per 8-k group, four groups per tile):

    group 0   wait for fragment set 0 | ds_read set 1 <- (b, k-group 1) | G MFMAs on set 0, between them: the cursor arithmetic
              (SALU), per gathered row the validity test + `buffer_load_dwordx4 ... lds` of tile t+1, then the weight rows
    group 1   wait set 1 | ds_read set 0 <- (b, k-group 2) | G MFMAs on set 1
    group 2   wait set 0 | ds_read set 1 <- (b, k-group 3) | G MFMAs on set 0
    group 3   wait set 1 (= every read of buffer b has returned) | G/4 MFMAs | s_waitcnt vmcnt(0) + s_barrier (tile t+1 has
              landed for everyone, buffer b is free) | ds_read set 0 <- (b^1, k-group 0) | the other 3G/4 MFMAs on set 1

Every fragment read is issued a whole group (G x 64 cycles) before its first use, so its s_waitcnt never stalls; hipcc's own
schedule of the same loop issues the reads behind the group's MFMAs, two MFMAs before their use (profiles/README.md r03).
The MFMA order per accumulator is the C++ loop's (k-group, then j = 0..3), so the results are bit-identical to it.

Registers: fragments live in v[FR .. FR+31] (named in the clobber list: inline-asm operands cannot address the single
registers of a 128-bit tuple, which the MFMA operands are); everything else is an operand.
"""
import os
import sys

FR = 96                       # first fragment register: 8 tuples of 4


def frag(s, name, nb_count):
    """register tuple base of fragment `name` (A0, A1, B0, B1) of set s"""
    idx = {"A0": 0, "A1": 4, "B0": 8, "B1": 12}[name]
    return FR + 16 * s + idx


def tup(r):
    return f"v[{r}:{r + 3}]"


class Gen:
    def __init__(self, BN, BM=128, WM=2, WN=2):
        """conv_mfma_kernel<BM, BN, WM, WN, true, true>: 128 x 128 and 128 x 64 (2 x 2 waves), and since round 5 the 64 x 128 tile of the
        one-sample launches (1 x 4 waves: every wave multiplies all 64 rows by its own 32 columns -- the 128 x 64 tile's wave shape
        with A and B swapped in size: two row passes of the gather, four passes of weight rows)."""
        self.BM, self.BN = BM, BN
        self.MB, self.NB = BM // (32 * WM), BN // (32 * WN)
        assert self.MB == 2 and self.NB in (1, 2)
        self.G = 4 * self.MB * self.NB
        self.A_ROWS = BM // 32
        self.B_PASS = BN // 32
        self.A_BUF = self.BM * 128           # bytes per A buffer
        self.B_BUF = self.BN * 128
        self.tag = str(BN) if BM == 128 else f"{BM}{BN}"
        self.out = []

    def e(self, s):
        self.out.append(s)

    # -- pieces
    def reads(self, buf, q, s):
        a, b = buf * self.A_BUF, buf * self.B_BUF
        r = [f"ds_read_b128 {tup(frag(s, 'A0', 0))}, %[la{q}] offset:{a}",
             f"ds_read_b128 {tup(frag(s, 'B0', 0))}, %[lb{q}] offset:{b}",
             f"ds_read_b128 {tup(frag(s, 'A1', 0))}, %[la{q}] offset:{a + 4096}"]
        if self.NB == 2:
            r.append(f"ds_read_b128 {tup(frag(s, 'B1', 0))}, %[lb{q}] offset:{b + 4096}")
        return r

    def mfmas(self, s):
        m = []
        for jj in range(4):
            for mb in range(self.MB):
                for nb in range(self.NB):
                    a = frag(s, f"A{mb}", 0) + jj
                    b = frag(s, f"B{nb}", 0) + jj
                    m.append(f"v_mfma_f32_32x32x2_f32 %[c{mb}{nb}], v{a}, v{b}, %[c{mb}{nb}]")
        return m

    def cursor_chunks(self):
        c1 = ["s_lshl_b32 %[qseg], %[kc], 5",
              "s_mul_i32 %[t], %[sg], %[lstride]",
              "s_add_i32 %[qabs], %[t], %[qseg]",
              "s_mul_i32 %[t], %[ky], %[pitch]",
              "s_add_i32 %[t], %[t], %[qabs]",
              "s_lshl_b32 %[soff], %[t], 2",
              "s_mov_b32 %[kyc], %[ky]"]
        c2 = ["s_add_i32 %[kc], %[kc], 1",
              "s_cmp_eq_u32 %[kc], %[kps]",
              "s_cselect_b32 %[kc], 0, %[kc]",
              "s_addc_u32 %[sg], %[sg], 0",
              "s_cmp_eq_u32 %[sg], %[nseg]",
              "s_cselect_b32 %[sg], 0, %[sg]",
              "s_addc_u32 %[ky], %[ky], 0"]
        return [c1, c2]

    def row_chunk(self, j, nxt):
        vo = f"%[vo{j & 1}]"
        return [f"s_add_i32 m0, %[ma], {nxt * self.A_BUF + j * 4096}",
                f"v_mov_b32 {vo}, 0xc0000000",
                f"v_sub_u32 %[vt], %[qabs], %[lo{j}]",
                f"v_cmpx_gt_u32 vcc, %[span{j}], %[vt]",
                f"v_add_u32 %[vt], %[kyc], %[y{j}]",
                f"v_cmpx_gt_u32 vcc, %[hi], %[vt]",
                f"v_cmpx_lt_i32 vcc, %[qseg], %[w{j}]",
                f"v_mov_b32 {vo}, %[x{j}]",
                "s_mov_b64 exec, -1",
                f"buffer_load_dwordx4 {vo}, %[rin], %[soff] offen lds"]

    def b_chunk(self, jb, nxt):
        c = [f"s_add_i32 m0, %[mb], {nxt * self.B_BUF + jb * 4096}",
             f"s_add_i32 %[t], %[soffw], {jb * 4096}",
             f"buffer_load_dwordx4 %[wv], %[rwt], %[t] offen lds"]
        if jb == self.B_PASS - 1:
            c.append("s_add_i32 %[soffw], %[soffw], %[wstep]")
        return c

    def group(self, pre, mf, chunks_at, post=()):
        """pre: lines before the first MFMA; mf: MFMA lines; chunks_at: {slot: [lines]} issued after MFMA `slot`"""
        for l in pre:
            self.e(l)
        for i, m in enumerate(mf):
            self.e(m)
            for l in chunks_at.get(i, ()):
                self.e(l)
        for l in post:
            self.e(l)

    def body(self, b, exit_label, next_label):
        nxt = b ^ 1
        G = self.G
        chunks = self.cursor_chunks() + [self.row_chunk(j, nxt) for j in range(self.A_ROWS)] + [self.b_chunk(jb, nxt) for jb in range(self.B_PASS)]
        stride = 1 if len(chunks) >= G else max(1, (G - 1) // len(chunks))
        at = {}
        for i, c in enumerate(chunks):
            at.setdefault(min(i * stride, G - 2), []).extend(c)
        self.group(["s_waitcnt lgkmcnt(0)"] + self.reads(b, 1, 1), self.mfmas(0), at)
        self.group(["s_waitcnt lgkmcnt(0)"] + self.reads(b, 2, 0), self.mfmas(1), {})
        self.group(["s_waitcnt lgkmcnt(0)"] + self.reads(b, 3, 1), self.mfmas(0), {})
        k = G // 4
        at3 = {k - 1: ["s_waitcnt vmcnt(0)", "s_barrier"] + self.reads(nxt, 0, 0),
               k: ["s_sub_u32 %[n], %[n], 1"]}
        post = ["s_cmp_eq_u32 %[n], 0", f"s_cbranch_scc1 {exit_label}"]
        if next_label:
            post.append(f"s_branch {next_label}")
        self.group(["s_waitcnt lgkmcnt(0)"], self.mfmas(1), at3, post)

    def tail(self, b, end_label):
        self.group(["s_waitcnt lgkmcnt(0)"] + self.reads(b, 1, 1), self.mfmas(0), {})
        self.group(["s_waitcnt lgkmcnt(0)"] + self.reads(b, 2, 0), self.mfmas(1), {})
        self.group(["s_waitcnt lgkmcnt(0)"] + self.reads(b, 3, 1), self.mfmas(0), {})
        self.group(["s_waitcnt lgkmcnt(0)"], self.mfmas(1), {}, [f"s_branch {end_label}"] if end_label else [])

    def generate(self):
        L = lambda n: f".Lvk{self.tag}_{n}_%="
        self.e("s_nop 4")
        for l in self.reads(0, 0, 0):
            self.e(l)
        self.e("s_cmp_eq_u32 %[n], 0")
        self.e(f"s_cbranch_scc1 {L('tail0')}")
        self.e(L("body0") + ":")
        self.body(0, L("tail1"), None)
        self.e(L("body1") + ":")
        self.body(1, L("tail0"), L("body0"))
        self.e(L("tail0") + ":")
        self.tail(0, L("end"))
        self.e(L("tail1") + ":")
        self.tail(1, None)
        self.e(L("end") + ":")
        self.e("s_nop 15")          # the last MFMAs' results are read by the epilogue's v_accvgpr_read: 18 wait states (16-pass XDL write -> VALU read)
        self.e("s_nop 7")
        return self.out

    def clobbers(self):
        return [f"v{FR + i}" for i in range(32)]     # the B1 tuples of the 64-column shape stay unused: one list for both shapes


class GemmStreamGen(Gen):
    """The Winograd-domain GEMMs of one stage (16 positions, each a [tiles x C_in] . [C_in x C_out] product with its own weights) as
    STREAMS: a workgroup keeps its 128 x 64 output tile and walks through P consecutive positions, kps K-tiles each.

    A position's reduction is short (8 or 16 K-tiles), so as one workgroup per (position, tile) the prologue (first operands' memory
    latency) and the epilogue (LDS transpose + store burst) are 40 % of a workgroup's life.  Here the K-tile stream simply runs on
    across positions -- the operand cursor jumps by one position plane when a position's K-tiles are through, the packed weights of
    consecutive positions are contiguous anyway -- and at a position's end the 32 accumulators are copied to registers, the next
    position starts with srcC = 0 and the finished tile leaves as 32 four-byte stores per lane between the MFMAs of k-groups 1 and
    2 of the next position's first K-tile.  All rows and K chunks are valid by construction (the launcher checks), so the fetch is
    a bare `buffer_load_dwordx4 ... lds` per row with the cursor in the scalar offset.  kps is even, so every position starts on
    buffer 0 and ends on buffer 1: bodies first (buffer 0; plain or with the previous tile's stores), mid (1, 0), last (1), final
    (1, nothing left to fetch).  Accumulators are a[0:31] by name."""

    ACC = {"c00": "a[0:15]", "c10": "a[16:31]"}

    def __init__(self):
        Gen.__init__(self, 64)

    def mfmas(self, s, zero=False):
        m = []
        for jj in range(4):
            for mb in range(self.MB):
                acc = self.ACC[f"c{mb}0"]
                a = frag(s, f"A{mb}", 0) + jj
                b = frag(s, "B0", 0) + jj
                m.append(f"v_mfma_f32_32x32x2_f32 {acc}, v{a}, v{b}, {'0' if (zero and jj == 0) else acc}")
        return m

    def fetch_chunks(self, nxt):
        c = []
        for j in range(self.A_ROWS):
            c.append([f"s_add_i32 m0, %[ma], {nxt * self.A_BUF + j * 4096}", "s_nop 0",
                      f"buffer_load_dwordx4 %[x{j}], %[rin], %[soff] offen lds"])
        for jb in range(self.B_PASS):
            c.append([f"s_add_i32 m0, %[mb], {nxt * self.B_BUF + jb * 4096}", f"s_add_i32 %[t], %[soffw], {jb * 4096}",
                      "buffer_load_dwordx4 %[wv], %[rwt], %[t] offen lds"])
        # the cursor: next K-tile of this position, or the first of the next position's plane
        c.append(["s_add_i32 %[soffw], %[soffw], %[wstep]", "s_add_i32 %[soff], %[soff], 128", "s_add_i32 %[kc], %[kc], 1",
                  "s_cmp_eq_u32 %[kc], %[kps]", "s_cselect_b32 %[t], %[posjump], 0", "s_cselect_b32 %[kc], 0, %[kc]",
                  "s_add_i32 %[soff], %[soff], %[t]"])
        return c

    def store_chunk(self, i):
        mb, r = divmod(i, 16)
        row = mb * 32 + (r & 3) + 8 * (r >> 2)
        return [f"v_add_f32 %[o{i}], 0, %[o{i}]",                # the zero bias the implicit-GEMM epilogue adds: -0 becomes +0 there too
                f"s_mul_i32 %[t], %[cs4], {row}", "s_add_i32 %[t], %[t], %[obase]",
                f"buffer_store_dword %[o{i}], %[vout], %[dout], %[t] offen"]

    def flush(self):
        self.e("s_nop 15")
        self.e("s_nop 7")
        for i in range(32):
            self.e(f"v_accvgpr_read_b32 %[o{i}], a{i}")

    def body(self, b, role):
        nxt = b ^ 1
        G = self.G
        at0 = {}
        if role != "final":
            for i, c in enumerate(self.fetch_chunks(nxt)):
                at0.setdefault(i, []).extend(c)
        self.group(["s_waitcnt lgkmcnt(0)"] + self.reads(b, 1, 1), self.mfmas(0, zero=role in ("first", "first_s")), at0)
        st = [self.store_chunk(i) for i in range(32)] if role == "first_s" else []
        at1, at2 = {}, {}
        for i, c in enumerate(st[:16]):
            at1.setdefault(i // 2, []).extend(c)
        for i, c in enumerate(st[16:]):
            at2.setdefault(i // 2, []).extend(c)
        self.group(["s_waitcnt lgkmcnt(0)"] + self.reads(b, 2, 0), self.mfmas(1), at1)
        self.group(["s_waitcnt lgkmcnt(0)"] + self.reads(b, 3, 1), self.mfmas(0), at2)
        if role == "final":
            self.group(["s_waitcnt lgkmcnt(0)"], self.mfmas(1), {})
        else:
            k = G // 4
            # (a counted wait that lets the previous tile's 32 stores fly on -- vmcnt(32) in the K-tile that carries them -- measured
            # 1 % on these launches and rests on loads and stores completing in issue order: not taken)
            vm = 0
            self.group(["s_waitcnt lgkmcnt(0)"], self.mfmas(1), {k - 1: [f"s_waitcnt vmcnt({vm})", "s_barrier"] + self.reads(nxt, 0, 0)})

    def generate(self):
        L = lambda n: f".Lvgs_{n}_%="
        self.e("s_nop 4")
        for l in self.reads(0, 0, 0):
            self.e(l)
        self.body(0, "first")
        self.e(f"s_branch {L('mids')}")
        self.e(L("pos") + ":")
        self.body(0, "first_s")
        self.e("s_add_i32 %[obase], %[obase], %[ostep]")
        self.e(L("mids") + ":")
        self.e("s_mov_b32 %[n], %[npairs]")
        self.e(L("pair") + ":")
        self.body(1, "mid")
        self.body(0, "mid")
        self.e("s_sub_u32 %[n], %[n], 1")
        self.e("s_cmp_lg_u32 %[n], 0")
        self.e(f"s_cbranch_scc1 {L('pair')}")
        self.e("s_sub_u32 %[npos], %[npos], 1")
        self.e("s_cmp_eq_u32 %[npos], 0")
        self.e(f"s_cbranch_scc1 {L('final')}")
        self.body(1, "last")
        self.flush()
        self.e(f"s_branch {L('pos')}")
        self.e(L("final") + ":")
        self.body(1, "final")
        self.flush()
        return self.out

    def clobbers(self):
        return [f"v{FR + i}" for i in range(32)] + [f"a{i}" for i in range(32)]


class RowWinGen:
    """conv_rowwin_kernel<7, 2> (the network's first layer): KH filter rows x KPR K-tiles, wave tile 64 pixels x 32 channels.

    One K-tile = 32 MFMAs (16 k-steps x 2 pixel blocks) fed by 8 ds_read2_b64 (A, straight out of the input-row window in LDS)
    and 4 buffer_load_dwordx4 (B, the packed weights, global -> registers).  Two register sets each; tile t multiplies set t&1
    while the loads of tile t+1 fill the other one: B first (longest latency), then one A read per MFMA.  The window of filter
    row ky+1 is fetched by LDS-DMA during tile 0 of row ky (no staging registers, no ds_write); the row's only barrier sits
    four MFMAs into its last tile: by then every wave has all its reads of this row's window behind it and its share of the
    next window landed, and the A reads of the next row's first tile follow under the remaining 28 MFMAs."""
    A = (64, 96)          # A fragment sets: 16 registers per pixel block, two blocks
    B = (128, 144)        # B fragment sets
    BUF = 7 * 4096        # byte stride of the two window buffers

    def __init__(self, kpr=6, MB=2):
        self.kpr = kpr
        self.MB = MB
        self.NWIN = 7 if MB == 2 else 4            # 4 KB wave chunks per window
        if MB == 1:                               # 64-pixel tiles (wave tile 32 x 32): half the A registers, lower so that four waves fit a SIMD
            self.A, self.B, self.BUF = (48, 64), (80, 96), 4 * 4096
        assert kpr % 2 == 0
        self.out = []

    def e(self, s):
        self.out.append(s)

    def mfmas(self, P):
        m = []
        for s in range(16):
            for mb in range(self.MB):
                m.append(f"v_mfma_f32_32x32x2_f32 %[c{mb}], v{self.A[P] + 16 * mb + s}, v{self.B[P] + s}, %[c{mb}]")
        return m

    def b_loads(self, Q):
        r = [f"buffer_load_dwordx4 v[{self.B[Q] + 4 * q}:{self.B[Q] + 4 * q + 3}], %[vb], %[dwt], %[koff] offen offset:{32 * q}" for q in range(4)]
        r[-1] = [r[-1], "s_add_i32 %[koff], %[koff], %[kstride]"]
        return [x if isinstance(x, list) else [x] for x in r]

    def a_reads(self, Q, kc, nxt):
        r = []
        for i in range(4):
            for mb in range(self.MB):
                base = self.A[Q] + 16 * mb + 4 * i
                addr = f"%[a{mb}{'n' if nxt else 'c'}]"
                r.append([f"ds_read2_b64 v[{base}:{base + 3}], {addr} offset0:{16 * kc + 4 * i} offset1:{16 * kc + 4 * i + 2}"])
        return r

    def dma(self):
        c = [["s_cmp_lt_u32 %[iy], %[hi]", "s_cselect_b64 %[mask], -1, 0"]]
        for j in range(self.NWIN):
            vt = f"%[vt{j & 1}]"
            c.append([f"s_add_i32 m0, %[mn], {j * 4096}",
                      f"v_cndmask_b32_e64 {vt}, -2.0, %[w{j}], %[mask]",
                      f"buffer_load_dwordx4 {vt}, %[din], %[soff] offen lds"])
        c.append(["s_add_i32 %[iy], %[iy], 1", "s_add_i32 %[soff], %[soff], %[rowbytes]"])
        return c

    def tile(self, P, wait_vm, chunks, barrier_after=None):
        self.e(f"s_waitcnt vmcnt({wait_vm}) lgkmcnt(0)")
        first = 0 if barrier_after is None else barrier_after + 1
        n = 16 * self.MB
        per = -(-len(chunks) // (n - first))          # chunks per MFMA slot (1 for the 128-pixel form)
        for i, m in enumerate(self.mfmas(P)):
            self.e(m)
            if barrier_after is not None and i == barrier_after:
                self.e("s_barrier")
            k = i - first
            for c in chunks[k * per:(k + 1) * per] if k >= 0 else ():
                for l in c:
                    self.e(l)

    def row(self, more):
        n = self.kpr
        for t in range(n):
            P, Q = t & 1, (t & 1) ^ 1
            last = t == n - 1
            if last and not more:
                self.tile(P, 0, [])
                continue
            chunks = self.b_loads(Q) + self.a_reads(Q, 0 if last else t + 1, last)
            if t == 0 and more:
                chunks += self.dma()
            self.tile(P, self.NWIN if (t == 1 and more) else 0, chunks, barrier_after=(3 if self.MB == 2 else 1) if last else None)

    def swap(self):
        self.e("v_swap_b32 %[a0c], %[a0n]")
        if self.MB == 2:
            self.e("v_swap_b32 %[a1c], %[a1n]")
        self.e("s_xor_b32 %[mn], %[mn], %[mx]")          # mx = m_cur ^ m_nxt: toggles between the two buffers' DMA bases

    def generate(self):
        L = lambda n: f".Lvrw{self.MB}_{n}_%="
        self.e("s_nop 4")
        for c in self.b_loads(0) + self.a_reads(0, 0, False):
            for l in c:
                self.e(l)
        self.e("s_cmp_eq_u32 %[nrows], 0")
        self.e(f"s_cbranch_scc1 {L('last')}")
        self.e(L("more") + ":")
        self.row(True)
        self.swap()
        self.e("s_sub_u32 %[nrows], %[nrows], 1")
        self.e("s_cmp_lg_u32 %[nrows], 0")
        self.e(f"s_cbranch_scc1 {L('more')}")
        self.e(L("last") + ":")
        self.row(False)
        self.e("s_nop 15")
        self.e("s_nop 7")
        return self.out

    def clobbers(self):
        return [f"v{i}" for i in range(self.A[0], self.B[1] + 16)]


class RowWinStreamGen(RowWinGen):
    """The same kernel as a STREAM of tiles: a workgroup keeps its x tile and sample and walks down R output rows (S rows apart; the
    kernel uses S = 1, consecutive rows: interleaving a column's streams measured slower at the large shapes).

    The per-lane window offsets never change; the first window of the next tile is one more LDS-DMA with the row cursor moved by
    fwd = S * stride - 7 rows, fetched during the last filter row of the current tile ('bridge' row) like any other window; the weights' K-tile cursor wraps to 0 in that row's last K-tile.  When a tile's
    last MFMA has issued, its 32 accumulators are copied to registers (after the 18 wait states an XDL write needs), the next tile's
    first MFMAs take srcC = 0, and the 32 four-byte stores per lane (bias, leaky relu as max(v, slope * v), scalar offset per pixel
    row) are issued between the MFMAs of K-tiles 1..5 of the next tile's first filter row -- the window buffers never serve as a
    staging area, the stream has no seams.  The last tile's values are handed back to the kernel's C++ epilogue.
    Accumulators are a[0:31] by name (operands cannot address single accumulator registers)."""

    ACC = ("a[0:15]", "a[16:31]")

    def mfmas(self, P, zero=False):
        m = []
        for s in range(16):
            for mb in range(2):
                c = "0" if (zero and s == 0) else self.ACC[mb]
                m.append(f"v_mfma_f32_32x32x2_f32 {self.ACC[mb]}, v{self.A[P] + 16 * mb + s}, v{self.B[P] + s}, {c}")
        return m

    def tile(self, P, wait_vm, chunks, barrier_after=None, zero=False):
        self.e(f"s_waitcnt vmcnt({wait_vm}) lgkmcnt(0)")
        first = 0 if barrier_after is None else barrier_after + 1
        for i, m in enumerate(self.mfmas(P, zero)):
            self.e(m)
            if barrier_after is not None and i == barrier_after:
                self.e("s_barrier")
            k = i - first
            if 0 <= k < len(chunks):
                for l in chunks[k]:
                    self.e(l)
        assert len(chunks) <= 32 - first, (len(chunks), first)

    def store_chunk(self, i):
        mb, r = divmod(i, 16)
        px = mb * 32 + (r & 3) + 8 * (r >> 2)
        o = f"%[o{i}]"
        return [f"v_add_f32 {o}, {o}, %[bias]",
                f"v_mul_f32 %[vt0], %[slope], {o}",
                f"v_max_f32 {o}, {o}, %[vt0]",
                f"s_mul_i32 %[st], %[cs4], {px}",
                "s_add_i32 %[st], %[st], %[obase]",
                f"buffer_store_dword {o}, %[vout], %[dout], %[st] offen"]

    def flush(self):
        self.e("s_nop 15")
        self.e("s_nop 7")
        for i in range(32):
            self.e(f"v_accvgpr_read_b32 %[o{i}], a{i}")

    def row(self, kind):
        n = self.kpr
        stores = list(range(32))
        per = [0, 7, 7, 7, 7, 4]                                # deferred stores per K-tile of a tile's first filter row
        for t in range(n):
            P, Q = t & 1, (t & 1) ^ 1
            last = t == n - 1
            if last and kind == "final":
                self.tile(P, 0, [])
                continue
            bl = self.b_loads(Q)
            if last and kind == "bridge":
                bl[0] = ["s_mov_b32 %[koff], 0"] + bl[0]        # the next tile starts at the weights' first K-tile
            chunks = bl + self.a_reads(Q, 0 if last else t + 1, last)
            if t == 0 and kind != "final":
                d = self.dma()
                if kind == "bridge":                            # the cursor stands 7 rows below this tile's first input row; the next
                    d[0] = ["s_mul_i32 %[st], %[rowbytes], %[fwd]",       # tile's first one is 2 * (rows between the stream's tiles) below it
                            "s_add_i32 %[soff], %[soff], %[st]", "s_add_i32 %[iy], %[iy], %[fwd]"] + d[0]
                chunks += d
            if kind == "first_s" and per[t]:
                for _ in range(per[t]):
                    chunks.append(self.store_chunk(stores.pop(0)))
            self.tile(P, 7 if (t == 1 and kind != "final") else 0, chunks, barrier_after=3 if last else None,
                      zero=(t == 0 and kind in ("first", "first_s")))
        assert not (kind == "first_s" and stores)

    def swap(self):
        self.e("v_swap_b32 %[a0c], %[a0n]")
        self.e("v_swap_b32 %[a1c], %[a1n]")
        self.e("s_xor_b32 %[mn], %[mn], %[mx]")

    def generate(self):
        L = lambda n: f".Lvrs_{n}_%="
        self.e("s_nop 4")
        for c in self.b_loads(0) + self.a_reads(0, 0, False):
            for l in c:
                self.e(l)
        self.row("first")
        self.swap()
        self.e(L("tile") + ":")
        self.e("s_mov_b32 %[nrows], 5")
        self.e(L("more") + ":")
        self.row("more")
        self.swap()
        self.e("s_sub_u32 %[nrows], %[nrows], 1")
        self.e("s_cmp_lg_u32 %[nrows], 0")
        self.e(f"s_cbranch_scc1 {L('more')}")
        self.e("s_sub_u32 %[ntiles], %[ntiles], 1")
        self.e("s_cmp_eq_u32 %[ntiles], 0")
        self.e(f"s_cbranch_scc1 {L('final')}")
        self.row("bridge")
        self.flush()
        self.swap()
        self.row("first_s")
        self.swap()
        self.e("s_add_i32 %[obase], %[obase], %[ostep]")
        self.e(f"s_branch {L('tile')}")
        self.e(L("final") + ":")
        self.row("final")
        self.flush()
        return self.out

    def clobbers(self):
        return RowWinGen.clobbers(self) + [f"a{i}" for i in range(32)]


class WgradGen:
    """wgrad_mfma_kernel<BN> (filter gradient; BN = 128, 64): K-tile = 32 output pixels, 16 k-steps of MB x NB = 2 x (BN/64) MFMAs.

    hipcc issued the fragment reads of k-step ks+1 and waited for them (`s_waitcnt lgkmcnt(0)`) in front of the MFMAs of k-step ks:
    an LDS round trip every four MFMAs (0.73 of the pipe).  Here the 16 k-steps run as four groups of four on two register sets,
    the reads of group g+1 (ds_read2st64_b32: one instruction fetches the same fragment of two consecutive k-steps) issued at the
    start of group g; the fetch of tile t+1 (pixel-table entries loaded a tile ahead, validity by EXEC narrowing,
    `buffer_load_dwordx4 ... lds`) sits between the MFMAs of group 0, then the table entries of tile t+2 are requested; the
    barrier comes four MFMAs into group 3 behind `s_waitcnt vmcnt(4)` (the DMA has landed, the table loads may still fly)."""
    FR = 96            # fragment sets: 16 registers each (a0 x4, a1 x4, b0 x4, b1 x4)
    PT = 80            # pixel-table entries of the next tile: 4 x {x, y, z} at PT + 4 j

    def __init__(self, BN):
        self.BN = BN
        self.NB = BN // 64
        self.G = 4 * 2 * self.NB                 # MFMAs per group of four k-steps
        self.BP = 4 if BN == 128 else 2          # DMA passes of the B tile
        self.BR = 8 if BN == 128 else 16         # B rows per pass
        self.B_STAGE = 32 * BN * 4               # bytes per stage of the B ring
        self.out = []

    def e(self, s):
        self.out.append(s)

    def reads(self, buf, g, s):
        """fragment reads of k-steps 4g .. 4g+3 out of stage buf into set s"""
        base = self.FR + 16 * s
        r = []
        a_u, b_u = buf * 64, buf * (self.B_STAGE // 256)
        bstep = (2 * self.BN * 4) // 256          # 256-byte units per k-step in the B tile
        for half in range(2):
            ks = 4 * g + 2 * half
            r.append(f"ds_read2st64_b32 v[{base + 2 * half}:{base + 2 * half + 1}], %[ra0] offset0:{a_u + ks * 4} offset1:{a_u + (ks + 1) * 4}")
            r.append(f"ds_read2st64_b32 v[{base + 8 + 2 * half}:{base + 8 + 2 * half + 1}], %[rb0] offset0:{b_u + ks * bstep} offset1:{b_u + (ks + 1) * bstep}")
            r.append(f"ds_read2st64_b32 v[{base + 4 + 2 * half}:{base + 4 + 2 * half + 1}], %[ra1] offset0:{a_u + ks * 4} offset1:{a_u + (ks + 1) * 4}")
            if self.NB == 2:
                r.append(f"ds_read2st64_b32 v[{base + 12 + 2 * half}:{base + 12 + 2 * half + 1}], %[rb1] offset0:{b_u + ks * bstep} offset1:{b_u + (ks + 1) * bstep}")
        return r

    def mfmas(self, s):
        base = self.FR + 16 * s
        m = []
        for i in range(4):
            for mb in range(2):
                for nb in range(self.NB):
                    m.append(f"v_mfma_f32_32x32x2_f32 %[c{mb}{nb}], v{base + 4 * mb + i}, v{base + 8 + 4 * nb + i}, %[c{mb}{nb}]")
        return m

    def fetch_chunks(self, nxt):
        c = []
        for j in range(4):                        # A: 8 pixel rows per pass
            vo = f"%[vo{j & 1}]"
            pt = self.PT + 4 * j
            c.append([f"s_add_i32 m0, %[ma], {nxt * 16384 + j * 4096}",
                      f"v_mov_b32 {vo}, 0xc0000000",
                      "s_mov_b64 exec, %[mok]",
                      f"v_add_u32 %[vt], {8 * j}, %[vk]",
                      "v_cmpx_gt_u32 vcc, %[kk], %[vt]",
                      f"v_add_u32 %[vt], v{pt + 1}, %[vky]",
                      "v_cmpx_gt_u32 vcc, %[hi], %[vt]",
                      f"v_add_u32 %[vt], v{pt + 2}, %[vkx]",
                      "v_cmpx_gt_u32 vcc, %[wi], %[vt]",
                      f"v_add_lshl_u32 {vo}, v{pt}, %[vad], 2",
                      "s_mov_b64 exec, -1",
                      f"buffer_load_dwordx4 {vo}, %[rx], 0 offen lds"])
        for j in range(self.BP):
            vo = f"%[vo{j & 1}]"
            ch = [f"s_add_i32 m0, %[mb], {nxt * self.B_STAGE + j * 4096}",
                  f"v_mov_b32 {vo}, 0xc0000000",
                  "s_mov_b64 exec, %[nok]",
                  f"v_add_u32 %[vt], {self.BR * j}, %[vkb]",
                  "v_cmpx_gt_u32 vcc, %[kk], %[vt]"]
            if j == 0:
                ch.append(f"v_mov_b32 {vo}, %[vbo]")
            else:
                ch += [f"s_mul_i32 %[t], %[brow], {j}", f"v_add_u32 {vo}, %[t], %[vbo]"]
            ch += ["s_mov_b64 exec, -1", f"buffer_load_dwordx4 {vo}, %[rg], 0 offen lds"]
            c.append(ch)
        # the table entries of the tile after the next one, then the cursors move on
        for j in range(4):
            pt = self.PT + 4 * j
            c.append([f"v_add_u32 %[vt], {32 + 8 * j}, %[vk]",
                      "v_min_u32 %[vt], %[km1], %[vt]",
                      "v_lshlrev_b32 %[vt], 4, %[vt]",
                      f"global_load_dwordx3 v[{pt}:{pt + 2}], %[vt], %[ptab]"])
        c.append(["v_add_u32 %[vk], 32, %[vk]", "v_add_u32 %[vkb], 32, %[vkb]", "v_add_u32 %[vbo], %[btile], %[vbo]"])
        return c

    def group(self, pre, mf, chunks_at, post=()):
        for l in pre:
            self.e(l)
        for i, m in enumerate(mf):
            self.e(m)
            for l in chunks_at.get(i, ()):
                self.e(l)
        for l in post:
            self.e(l)

    def body(self, b, exit_label, next_label):
        nxt = b ^ 1
        G = self.G
        chunks = self.fetch_chunks(nxt)
        at = {}
        if len(chunks) <= G - 1:
            for i, c in enumerate(chunks):
                at.setdefault(i, []).extend(c)
        else:                                     # the 64-column shape: 8 MFMAs per group, 11 chunks -- two per slot where needed
            for i, c in enumerate(chunks):
                at.setdefault(i * (G - 1) // len(chunks), []).extend(c)
        # group 0 waits for the fragments AND for the table entries this tile's fetch reads
        self.group(["s_waitcnt vmcnt(0) lgkmcnt(0)"] + self.reads(b, 1, 1), self.mfmas(0), at)
        self.group(["s_waitcnt lgkmcnt(0)"] + self.reads(b, 2, 0), self.mfmas(1), {})
        self.group(["s_waitcnt lgkmcnt(0)"] + self.reads(b, 3, 1), self.mfmas(0), {})
        k = max(G // 4, 2)
        at3 = {k - 1: ["s_waitcnt vmcnt(4)", "s_barrier"] + self.reads(nxt, 0, 0), k: ["s_sub_u32 %[n], %[n], 1"]}
        post = ["s_cmp_eq_u32 %[n], 0", f"s_cbranch_scc1 {exit_label}"]
        if next_label:
            post.append(f"s_branch {next_label}")
        self.group(["s_waitcnt lgkmcnt(0)"], self.mfmas(1), at3, post)

    def tail(self, b, end_label):
        self.group(["s_waitcnt lgkmcnt(0)"] + self.reads(b, 1, 1), self.mfmas(0), {})
        self.group(["s_waitcnt lgkmcnt(0)"] + self.reads(b, 2, 0), self.mfmas(1), {})
        self.group(["s_waitcnt lgkmcnt(0)"] + self.reads(b, 3, 1), self.mfmas(0), {})
        self.group(["s_waitcnt lgkmcnt(0)"], self.mfmas(1), {}, [f"s_branch {end_label}"] if end_label else [])

    def generate(self):
        L = lambda n: f".Lvwg{self.BN}_{n}_%="
        self.e("s_nop 4")
        for j in range(4):                        # the table entries of tile kt0 + 1, fetched by the C++ prologue, into their fixed registers
            for c, nm in enumerate("xyz"):
                self.e(f"v_mov_b32 v{self.PT + 4 * j + c}, %[p{nm}{j}]")
        for l in self.reads(0, 0, 0):
            self.e(l)
        self.e("s_cmp_eq_u32 %[n], 0")
        self.e(f"s_cbranch_scc1 {L('tail0')}")
        self.e(L("body0") + ":")
        self.body(0, L("tail1"), None)
        self.e(L("body1") + ":")
        self.body(1, L("tail0"), L("body0"))
        self.e(L("tail0") + ":")
        self.tail(0, L("end"))
        self.e(L("tail1") + ":")
        self.tail(1, None)
        self.e(L("end") + ":")
        self.e("s_waitcnt vmcnt(0)")              # table loads of a tile that will not come
        self.e("s_nop 15")
        self.e("s_nop 7")
        return self.out

    def clobbers(self):
        return [f"v{i}" for i in range(self.FR, self.FR + 32)] + [f"v{self.PT + 4 * j + c}" for j in range(4) for c in range(3)]


def render():
    o = ["// GENERATED by tools/gen_conv_kloop.py -- do not edit; see that file for the schedule this encodes.",
         "// clang-format off"]
    for BN in (128, 64):
        g = Gen(BN)
        lines = g.generate()
        o.append(f"#define VSTAB_KLOOP_ASM_128x{BN} \\")
        for l in lines:
            o.append(f'    "{l}\\n" \\')
        o.append("    \"\"")
        if BN == 128:
            o.append("#define VSTAB_KLOOP_CLOBBERS " + ", ".join(f'"{c}"' for c in g.clobbers()))
    g = Gen(128, BM=64, WM=1, WN=4)
    o.append("#define VSTAB_KLOOP_ASM_64x128 \\")
    for l in g.generate():
        o.append(f'    "{l}\\n" \\')
    o.append("    \"\"")
    r1 = RowWinGen(6, MB=1)
    o.append("#define VSTAB_ROWWIN1_ASM_KPR6 \\")
    for l in r1.generate():
        o.append(f'    "{l}\\n" \\')
    o.append("    \"\"")
    o.append("#define VSTAB_ROWWIN1_CLOBBERS " + ", ".join(f'"{c}"' for c in r1.clobbers()))
    o.append("#define VSTAB_ROWWIN1_BUF_BYTES " + str(r1.BUF))
    r = RowWinGen(6)
    o.append("#define VSTAB_ROWWIN_ASM_KPR6 \\")
    for l in r.generate():
        o.append(f'    "{l}\\n" \\')
    o.append("    \"\"")
    o.append("#define VSTAB_ROWWIN_CLOBBERS " + ", ".join(f'"{c}"' for c in r.clobbers()))
    o.append("#define VSTAB_ROWWIN_BUF_BYTES " + str(RowWinGen.BUF))
    gs = GemmStreamGen()
    o.append("#define VSTAB_GEMM_STREAM_ASM \\")
    for l in gs.generate():
        o.append(f'    "{l}\\n" \\')
    o.append("    \"\"")
    o.append("#define VSTAB_GEMM_STREAM_CLOBBERS " + ", ".join(f'"{c}"' for c in gs.clobbers()))
    rs = RowWinStreamGen(6)
    o.append("#define VSTAB_ROWWIN_STREAM_ASM_KPR6 \\")
    for l in rs.generate():
        o.append(f'    "{l}\\n" \\')
    o.append("    \"\"")
    o.append("#define VSTAB_ROWWIN_STREAM_CLOBBERS " + ", ".join(f'"{c}"' for c in rs.clobbers()))
    for BN in (128, 64):
        w = WgradGen(BN)
        o.append(f"#define VSTAB_WGRAD_ASM_{BN} \\")
        for l in w.generate():
            o.append(f'    "{l}\\n" \\')
        o.append("    \"\"")
        if BN == 128:
            o.append("#define VSTAB_WGRAD_CLOBBERS " + ", ".join(f'"{c}"' for c in w.clobbers()))
    o.append("// clang-format on")
    return "\n".join(o) + "\n"


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "coupe", "optical_flow_based_deep_video_stabilization_amd", "csrc",
                        "conv_kloop_gfx950.inc")
    text = render()
    if len(sys.argv) > 1 and sys.argv[1] == "--print":
        sys.stdout.write(text)
    else:
        with open(path, "w") as f:
            f.write(text)
        print("wrote", os.path.normpath(path), len(text.splitlines()), "lines")
