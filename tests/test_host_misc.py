"""More host-side logic checked without a GPU: batch chunking under the 2 GiB tensor limit, VGG16 / NLDF shape
and workspace planning, the row-window weight layout of the first layer, size rule vs the oracle."""
import ctypes as C

import numpy as np
import pytest

from coupe.optical_flow_based_deep_video_stabilization_amd import _lib, netspec
from oracle import vstab_oracle as vo

L = _lib.lib()


def test_workspace_is_sized_for_one_chunk():
    one = L.vstab_workspace_bytes(1, 1080, 1920, 27)
    assert one > 0
    # a 1080p sample's input is 224 MB -> at most 9 fit under 2 GiB; larger batches reuse the chunk workspace
    b9, b10, b16, b125 = (L.vstab_workspace_bytes(b, 1080, 1920, 27) for b in (9, 10, 16, 125))
    assert b10 == b16 == b125 == b9 and b9 > one
    assert L.vstab_workspace_bytes(0, 512, 512, 27) == 0 and L.vstab_workspace_bytes(1, 2, 2, 27) == 0


@pytest.mark.parametrize("H,W", [(256, 256), (384, 512), (720, 1280), (1080, 1920), (70, 90)])
def test_level_sizes_match_oracle_and_netspec(H, W):
    buf = (C.c_int32 * 20)()
    assert L.vstab_level_sizes(H, W, buf) == 0
    got = [(buf[2 * i], buf[2 * i + 1]) for i in range(10)]
    assert got == vo.level_sizes(H, W) == list(netspec.sizes_for(H, W).enc)


def test_vgg16_shapes_and_workspace():
    hwc = (C.c_int32 * 54)()
    assert L.vstab_vgg16_shapes(1080, 1920, hwc) == 0
    shapes = [(hwc[3 * i], hwc[3 * i + 1], hwc[3 * i + 2]) for i in range(18)]
    assert shapes[0] == (1080, 1920, 64) and shapes[2] == (540, 960, 64) and shapes[-1] == (34, 60, 512)
    assert shapes[9] == (135, 240, 256) and shapes[13] == (68, 120, 512)       # SAME pooling rounds up: 135 -> 68
    assert L.vstab_vgg16_workspace_bytes(16, 1080, 1920) >= 256
    assert L.vstab_vgg16_shapes(0, 10, hwc) != 0
    assert L.vstab_nldf_workspace_bytes(2) > L.vstab_nldf_workspace_bytes(1) > 0 and L.vstab_nldf_workspace_bytes(0) == 0


def test_chunks_are_equalised():
    # 32 samples at 720p: max chunk 21 -> two chunks of 16 (same plan, bit-identical results); the plan of a
    # 16-sample batch is what both chunks use
    buf21, buf16 = (C.c_int32 * 64)(), (C.c_int32 * 64)()
    assert L.vstab_host_layer_plan(21, 720, 1280, 27, 4, buf21, 64) > 0
    assert L.vstab_host_layer_plan(22, 720, 1280, 27, 4, buf21, 64) < 0          # 22 no longer fits one chunk
    assert L.vstab_host_layer_plan(16, 720, 1280, 27, 4, buf16, 64) > 0 and buf16[0] == 16


def test_errors_without_context():
    assert L.vstab_flownets_forward(None, None, 1, 64, 64, 27, None, None, None, None, None, None, 0, None) < 0
    assert b"ctx" in L.vstab_last_error(None)
    assert L.vstab_warp_flow(None, None, None, 1, 4, 4, 3, None) < 0
    assert L.vstab_flow_box_blur(None, 1, 4, 4, 3, None, None, None) < 0
    assert L.vstab_vec2mtrx(None, 1, 8, 4, None, None) < 0
    # round-4 entry points: no context / NULL buffers are error codes, never a crash
    assert L.vstab_set_plan_batch(None, 8) == -6 and L.vstab_set_plan_flags(None, 1) == -6
    assert L.vstab_transform_image(None, 1, 4, 4, 3, None, None, None, 4, 4, None) == -6
    assert L.vstab_workspace_bytes_ctx(None, 8, 512, 512, 27) == L.vstab_workspace_bytes(8, 512, 512, 27)      # no context = unpinned
    assert L.vstab_workspace_bytes_ctx(None, 0, 512, 512, 27) == 0


@pytest.mark.parametrize("grid", [(256, 1, 1), (256, 2, 1), (64, 4, 8), (1024, 1, 4), (13, 1, 19), (11, 3, 5), (7, 1, 1), (1, 1, 1),
                                  (256, 2, 16), (384, 2, 1)])
def test_xcd_remap_is_a_bijection_and_bands_the_tiles(grid):
    gx, gy, gz = grid
    T = gx * gy * gz
    xyz = (C.c_int32 * 3)()
    seen, per_label = set(), {c: [] for c in range(8)}
    for lin in range(T):
        assert L.vstab_host_xcd_remap(gx, gy, gz, lin, xyz) == 0
        x, y, z = xyz[0], xyz[1], xyz[2]
        assert 0 <= x < gx and 0 <= y < gy and 0 <= z < gz
        seen.add((x, y, z))
        per_label[lin % 8].append(y + gy * (z + gz * x))             # position in the (y, z, x) tile order
    assert len(seen) == T                                            # bijective for any grid size
    # workgroups that share an XCD (same lin % 8) get one CONTIGUOUS, ascending range of that order; the ranges tile [0, T)
    ranges = []
    for c in range(8):
        p = per_label[c]
        if p:
            assert p == list(range(p[0], p[0] + len(p)))
            ranges.append((p[0], p[0] + len(p)))
    ranges.sort()
    assert ranges[0][0] == 0 and ranges[-1][1] == T and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    assert L.vstab_host_xcd_remap(4, 4, 4, 64, xyz) != 0                # out of range


def test_instrumentation_entry_points_without_gpu():
    """roctx ranges resolve their library lazily; the HBM-side profile can be switched and read with nothing recorded."""
    import ctypes as C
    from coupe.optical_flow_based_deep_video_stabilization_amd import _lib
    L = _lib.lib()
    assert L.vstab_trace_ranges(1) in (0, -6)          # VSTAB_OK, or VSTAB_E_STATE when no roctx library is installed
    assert L.vstab_trace_ranges(0) == 0
    assert L.vstab_hbm_profile_enable(1) == 0 and L.vstab_hbm_profile_enable(0) == 0
    ms, n, by = C.c_double(-1), C.c_int(-1), C.c_double(-1)
    assert L.vstab_hbm_profile_read(2, C.byref(ms), C.byref(n), C.byref(by)) == 0
    assert (ms.value, n.value, by.value) == (0.0, 0, 0.0)
    assert L.vstab_hbm_profile_read(3, C.byref(ms), C.byref(n), C.byref(by)) == 0 and L.vstab_hbm_profile_read(7, C.byref(ms), C.byref(n), C.byref(by)) < 0
    assert L.vstab_hbm_profile_enable(5) < 0
