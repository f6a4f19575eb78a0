#!/usr/bin/env python3
"""Does the ORDER of instruction classes matter for VALU issue?  The probe's composition (one trip of the level-0 record loop) issued
interleaved in small blocks (as bench.py's probe), grouped per point, grouped per trip, and with single instructions alternating, at
4 / 5 / 8 waves per SIMD.  Experiment build of the library (make -C dense-visual-odometry_amd/csrc experiments)."""
import ctypes as C
import os
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "dense-visual-odometry_amd"))
os.environ.setdefault("DVO_HIP_LIB", str(ROOT / "ab_libs" / "libdvo_hip_exp.so"))
import numpy as np
import dvo_amd
from dvo_amd import _cabi

with dvo_amd.HipContext(max_w=64, max_h=48, max_levels=1, n_slots=2, max_pairs=1) as ctx:
    lib = _cabi.lib()
    for waves in (4, 5, 8):
        out = np.zeros(72, dtype=np.float64)
        comp = np.zeros(4, dtype=np.int32)
        rc = lib.dvo_debug_valu_rate(ctx._h, waves, 4000, comp.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.POINTER(C.c_double)))
        assert rc == 0
        names = ["grouped per point (bench.py probe)", "fast alone", "bank-conflict fmac alone", "slow alone", "v_rcp_f32 alone", "blocks of 8/4/4", "grouped per trip",
                 "single instructions alternating (2 fast : 1 slow)",
                 "256 fast, 16 v_rcp each followed by fast", "256 fast, then 16 v_rcp in a row", "256 fast + 64 slow, v_rcp followed by slow", "256 fast + 64 slow, then 16 v_rcp in a row",
                 "256 fast, v_rcp + s_nop then fast", "256 fast + 64 slow, no v_rcp",
                 "256 fast, v_rcp + s_nop 1", "256 fast, v_rcp + s_nop 3", "256 fast, v_rcp + s_nop 7", "256 fast, v_rcp + s_mov", "256 fast, s_nop + v_rcp + s_nop",
                 "256 fast, 8 x (v_rcp v_rcp s_nop)", "256 fast, v_rcp + v_nop", "256 fast, v_rcp + one v_cvt",
                 "per point, slow spread between the FMAs (4 : 10)", "per point, slow AND conflict FMAs spread",
                 "v_cvt_f32_i32 alone", "v_bfe_i32 alone", "v_ashrrev_i32 alone", "v_cvt_f32_ubyte0 alone", "v_and_or_b32 alone", "v_bfi_b32 alone", "v_lshlrev_b32 alone",
                 "v_fract_f32 alone", "v_cmp + v_cndmask alone", "trip with the texel unpack by mask-and-merge (16 slow -> simple per point)",
                 "-", "-", "-", "-", "-", "-",
                 "v_lshlrev_b32 (rotating destinations)", "v_lshrrev_b32", "v_ashrrev_i32", "v_and_b32 literal", "v_and_b32", "v_or_b32", "v_mov_b32", "v_mov_b64", "v_add_u32", "v_sub_f32",
                 "v_mul_f32", "v_cvt_f32_u32", "v_cvt_u32_f32", "v_cvt_f32_ubyte3", "v_mad_u32_u24", "v_perm_b32", "v_max_f32", "v_med3_f32", "v_fma_f32", "v_bfe_u32",
                 "v_cvt_f32_i32", "v_fract_f32", "v_cvt_f32_f16", "v_cvt_f32_ubyte1",
                 "v_fmac_f32 acc, a, a (square)", "v_fmac_f32_dpp acc, a, a (identity quad_perm)", "v_mul_f32 d, a, a", "v_fmac_f32 acc, a, b (one bank)", "v_fmac_f32_dpp acc, a, b (one bank)",
                 "v_fmac_f32 acc, a, b (two banks)", "v_fmac_f32_dpp acc, a, b (two banks)", "v_mul_f32_sdwa d, a, a"]
        counts = [549, 512, 512, 512, 128, 549, 548, 456, 272, 272, 336, 336, 272, 320, 272, 272, 272, 272, 272, 272, 288, 288, 548, 548, 512, 512, 512, 512, 512, 512, 512, 512, 512, 541] + [512] * 38
        counts[0] = counts[5] = int(comp.sum())             # modes 0 and 5 issue the composition the library was built with
        print("%d waves per SIMD, composition %s" % (waves, [int(v) for v in comp]))
        for n, v, cnt in zip(names, out, counts):
            if n == "-":
                continue
            print("    %-52s %.3f ns per wave-instruction per SIMD   (%6.1f ns per iteration of %d)" % (n, v, v * cnt, cnt))
        f, c, s, r = comp
        print("    class-priced sum for the composition: %.3f ns" % ((f * out[1] + c * out[2] + s * out[3] + r * out[4]) / (f + c + s + r)))
