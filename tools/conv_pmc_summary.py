#!/usr/bin/env python3
"""Per-layer summary of tools/gpu_conv_pmc.sh's passes:  python tools/conv_pmc_summary.py gpurun_out/conv_pmc2 [batch]
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES); a v_mfma_f32_32x32x16_f16 occupies the SIMD's
matrix pipe for 32 cycles (MI355X_MICROARCH.md), i.e. 100 % = 1024 FLOP / clk / SIMD = the 2.5 PFLOP/s dense f16 peak at 2.4 GHz.
HBM bytes: FETCH_SIZE (KiB, doubled: the gfx950 wide-read correction of the same guide) + WRITE_SIZE (KiB)."""
import os
import sqlite3
import sys

src = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
SHAPES = [("bev conv1_1", 608, 608, 9, 64), ("rgb conv1_1", 375, 1242, 3, 64), ("bev conv1_2", 608, 608, 64, 64),
          ("bev conv2_2", 304, 304, 128, 128), ("bev conv3_2", 152, 152, 256, 256), ("bev conv4_1", 76, 76, 256, 512),
          ("bev conv4_2", 76, 76, 512, 512), ("rgb conv1_2", 375, 1242, 64, 64), ("rgb conv2_2", 187, 621, 128, 128),
          ("rgb conv3_2", 93, 310, 256, 256), ("rgb conv4_2", 46, 155, 512, 512)]


def grid(H, W, cout):
    bm, bn = (128, 128) if cout % 128 == 0 else (256, 64)
    mt = (B * H * W + bm - 1) // bm
    return (mt + 7) // 8 * 8 * (cout // bn) * 256


def load(sub):
    db = os.path.join(src, sub, "r_results.db")
    out = {}
    if os.path.exists(db):
        c = sqlite3.connect(db)
        for g, name, v, d in c.execute("select grid_size, counter_name, avg(value), avg(duration) from counters_collection where "
                                       "kernel_name like '%conv3x3%' group by grid_size, counter_name"):
            out.setdefault(g, {})[name] = v
            out[g]["_dur_ns"] = d
    return out


m, l, f, w = load("mfma"), load("lds"), load("fetch"), load("write")
print("%-12s %9s %8s %9s %10s %10s %9s %9s" % ("layer", "GFLOP", "MFMA %", "LDS cnfl%", "HBM rd MB", "HBM wr MB", "alg rd MB", "alg wr MB"))
for name, H, W, cin, cout in SHAPES:
    g = grid(H, W, cout)
    fl = 2.0 * B * H * W * cout * 9 * cin
    mm, ll = m.get(g, {}), l.get(g, {})
    util = 100 * mm.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(4 * mm.get("SQ_BUSY_CU_CYCLES", 1), 1)
    cn = 100 * ll.get("SQ_LDS_BANK_CONFLICT", 0) / max(ll.get("SQ_LDS_IDX_ACTIVE", 1), 1)
    rd = 2 * f.get(g, {}).get("FETCH_SIZE", 0) * 1024 / 1e6
    wr = w.get(g, {}).get("WRITE_SIZE", 0) * 1024 / 1e6
    cin_p = 16 if cin < 16 else cin
    alg_rd = (B * (H + 2) * (W + 2) * cin_p * 2 + cout * 9 * cin_p * 2) / 1e6
    alg_wr = B * H * W * cout * 2 / 1e6
    print("%-12s %9.1f %7.1f%% %8.1f%% %10.1f %10.1f %9.1f %9.1f" % (name, fl / 1e9, util, cn, rd, wr, alg_rd, alg_wr))
