"""Host logic of the grouped GEMM's launch-scheme choice (llm_bci_amd/csrc/gemm_streamk.hip) through nbci_debug_gemm_grouped_plan: no
launch, no device access, so it runs without a GPU. What is pinned: a layer's four weight gradients at the bench batch are dealt out over
512 workgroups in contiguous runs (3/4 tile per workgroup), groups whose tile/slot ratio has a large denominator take the owner + helper
scheme, a K that is not a multiple of 64 still qualifies, full or tiny groups keep one workgroup per tile, mixed layouts fall back to one
launch per problem. The reference has no counterpart (the matmuls inside loss.backward(), models/trainer.py:339)."""
import ctypes as C

import pytest

from llm_bci_amd import _lib as L

NBCI_BF16, NBCI_F32 = 1, 0


def _lib_or_skip():
    try:
        return L.lib()
    except L.NbciUnavailable as e:      # (the library is built by __graft_entry__.build(); hipcc cross-compiles without a GPU)
        pytest.skip(str(e))


def _operand(kmajor, ld):
    o = L.Operand()
    o.ptr, o.ld, o.kmajor = 0x100000, ld, 1 if kmajor else 0      # never dereferenced: alignment is all the planner looks at
    return o


def _plan(probs, ak=False, bk=False, mode=1, cus=256):
    l = _lib_or_skip()
    descs = (L.GemmDesc * len(probs))()
    for d, (M, N, K) in zip(descs, probs):
        d.M, d.N, d.K, d.in_dtype = M, N, K, NBCI_BF16
        d.A, d.B = _operand(ak, K if ak else M), _operand(bk, K if bk else N)
        d.C, d.ldc, d.c_dtype, d.batch, d.zdiv, d.splitk, d.alpha, d.beta = 0x200000, N, NBCI_F32, 1, 1, 1, 1.0, 1.0
    out = (C.c_int32 * 8)()
    L.check(l.nbci_debug_gemm_streamk(mode), "mode")
    L.check(l.nbci_set_available_cus(cus), "cus")
    try:
        L.check(l.nbci_debug_gemm_grouped_plan(descs, len(probs), out), "plan")
    finally:
        L.check(l.nbci_debug_gemm_streamk(1), "mode")
        L.check(l.nbci_set_available_cus(256), "cus")
    return dict(zip(("dealt", "scheme", "workgroups", "q", "rem", "slots", "tiles", "kt"), list(out)))


LAYER = [(3072, 1024, 9152), (1024, 1024, 9152), (1024, 1024, 9152), (1024, 1024, 9152)]


def test_layer_group_is_dealt_out_in_contiguous_runs():
    p = _plan(LAYER)
    assert p["dealt"] == 1 and p["scheme"] == 0 and p["workgroups"] == 512 and p["tiles"] == 384 and p["kt"] == 143 and p["slots"] == 512


def test_layer_group_other_schemes_on_request():
    a = _plan(LAYER, mode=3)      # owner + helper: owners do ceil(143 * 384 / 512) = 108 K tiles, helpers the remaining 35 of three tiles each
    assert (a["dealt"], a["scheme"], a["q"], a["rem"]) == (1, 1, 108, 35) and a["slots"] == 512 * 4
    b = _plan(LAYER, mode=4)      # six 8 x 8 blocks, one per XCD, helpers of three blocks on each of the other two XCDs
    assert (b["dealt"], b["scheme"], b["q"], b["rem"]) == (1, 2, 108, 35) and b["slots"] == 512 * 3
    assert _plan(LAYER, mode=0)["dealt"] == 0


def test_tile_to_slot_ratio_with_large_denominator_takes_the_aligned_scheme():
    p = _plan([(2304, 768, 24016), (768, 768, 24016), (2048, 768, 24016), (768, 2048, 24016)])   # 336 tiles: 21 / 32; K = 375 x 64 + 16
    assert p["dealt"] == 1 and p["scheme"] == 1 and p["tiles"] == 336 and p["kt"] == 375
    assert p["q"] == -(-375 * 336 // 512) and p["q"] + p["rem"] == 375


def test_full_or_tiny_groups_keep_one_workgroup_per_tile():
    assert _plan([(4096, 2048, 4096)])["dealt"] == 0                      # 512 tiles: nothing idle
    assert _plan([(256, 128, 1000), (128, 384, 1000)])["dealt"] == 0     # 7 tiles x 15 K tiles: not worth dealing out
    assert _plan([(1024, 1024, 9152), (1024, 1024, 9152)])["dealt"] == 1  # 128 tiles on 512 slots


def test_fewer_available_cus_shrink_the_launch():
    p = _plan(LAYER, cus=224)
    assert p["dealt"] == 1 and p["workgroups"] == 448


def test_mixed_layouts_are_launched_one_by_one():
    l = _lib_or_skip()
    descs = (L.GemmDesc * 2)()
    for i, d in enumerate(descs):
        d.M, d.N, d.K, d.in_dtype = 1024, 1024, 4096, NBCI_BF16
        d.A, d.B = _operand(i == 0, 4096 if i == 0 else 1024), _operand(False, 1024)
        d.C, d.ldc, d.c_dtype, d.batch, d.zdiv, d.splitk, d.alpha, d.beta = 0x200000, 1024, NBCI_F32, 1, 1, 1, 1.0, 0.0
    out = (C.c_int32 * 8)()
    L.check(l.nbci_debug_gemm_grouped_plan(descs, 2, out), "plan")
    assert out[0] == -1
