#!/usr/bin/env python3
"""Summarise the per-wavefront trace records an IMMESH_DEBUG=1 IMMESH_TRACE_FILE=<f> run leaves behind (reg_kernels.hip: DBG_* layout).
replay_fused_kernel: one record per wavefront of the LAST launch; residual_persistent_kernel: one per (pass, block) of the last scan.
Times: s_memrealtime ticks (100 MHz -> 10 ns) for start / end, shader cycles for the phases."""
import sys
import numpy as np

w = np.fromfile(sys.argv[1], dtype=np.uint64)
F_OFF, F_N = 64, 16384
R_OFF = F_OFF + F_N * 8
f = w[F_OFF:R_OFF].reshape(F_N, 8).astype(np.int64)
used = f[:, 0] > 0
f = f[used]
if len(f):
    f = f[f[:, 0] >= f[:, 0].max() - 30000]      # the LAST launch only (records of earlier launches with more wavefronts stay behind): 300 us window
    t0 = f[:, 0].min()
    start, end = (f[:, 0] - t0) * 0.01, (f[:, 1] - t0) * 0.01          # us
    flags = f[:, 7]
    state = (flags >> 16) & 0xFF; cnt = flags & 0xFF; nref = (flags >> 8) & 0xFF
    ph_all = np.ascontiguousarray(f[:, 2:7]).view(np.uint32).reshape(len(f), 10).astype(np.float64)
    print(f"replay_fused: {len(f)} wavefronts traced, span {end.max():.1f} us; exit-only {int((state == 0).sum())}, dropped {int((state == 1).sum())}, "
          f"done {int((state == 2).sum())} (with a fit {int(((state == 2) & (nref > 0)).sum())}), handed over {int((state == 3).sum())}")
    for name, sel in (("exit-only", state == 0), ("dropped", state == 1), ("done, no fit", (state == 2) & (nref == 0)), ("done, fit", (state == 2) & (nref > 0)), ("handed over", state == 3)):
        if sel.sum() == 0:
            continue
        d = end[sel] - start[sel]
        print(f"  {name:14s} n={int(sel.sum()):5d}  start p50/p99/max {np.percentile(start[sel], 50):6.1f} {np.percentile(start[sel], 99):6.1f} {start[sel].max():6.1f} us | "
              f"duration p50/p99/max {np.percentile(d, 50):6.2f} {np.percentile(d, 99):6.2f} {d.max():6.2f} us | end max {end[sel].max():6.1f} us | cnt max {int(cnt[sel].max())}")
        if name != "exit-only":
            ph = ph_all[sel]
            names = ["root", "node0", "chunks", "head", "list", "sort", "load", "decide", "commit", "plane"]
            print("     phases (cycles, mean / max): " + "  ".join(f"{n} {ph[:, i].mean():.0f}/{ph[:, i].max():.0f}" for i, n in enumerate(names)))
r = w[R_OFF:R_OFF + 8 * 512 * 8].reshape(8, 512, 8).astype(np.int64)
if (r[0, :, 0] > 0).any():
    t_last = r[:, :, 0].max()
    r = np.where((r[:, :, 0:1] >= t_last - 30000), r, 0)      # the last scan only
    t0 = r[0][r[0, :, 0] > 0][:, 0].min()
    print("residual_persistent (us since the first wavefront's start):")
    b0 = r[0][r[0, :, 0] > 0]
    if (b0[:, 4] > 0).any():
        ent = (b0[:, 4] - t0) * 0.01
        fin = (r[0, 0, 5] - t0) * 0.01
        print(f"  kernel entry (first wavefront of a block) min/max {ent.min():6.2f} {ent.max():6.2f}; block 0 finished (posterior out, ticket) {fin:6.2f}")
        b1 = r[1][: len(b0)]
        if len(sys.argv) > 2 and (b1[:, 5] >= 0).any():   # where each block ran: entry time, pass-0 partials out, XCC, SE, CU (HW_ID: cu_id bits 8-11, sh 12, se 13-15 on gfx9)
            out0 = (b0[:, 1] - t0) * 0.01
            print("  per block (entry us, pass-0 partials out us, xcc, se, cu): " + "  ".join(
                f"[{ent[i]:.1f} {out0[i]:.1f} x{int(b1[i, 5]) & 15} s{(int(b1[i, 4]) >> 13) & 7} c{(int(b1[i, 4]) >> 8) & 15}]" for i in np.argsort(ent)))
        if (b0[:, 7] > 0).any():
            e5 = (b0[1:, 5] - t0) * 0.01; e6 = (b0[:, 6] - t0) * 0.01; e7 = (b0[:, 7] - t0) * 0.01
            print(f"  epilogue: start (blocks > 0) min/max {e5.min():6.2f} {e5.max():6.2f}; points prepared p50/max {np.percentile(e6, 50):6.2f} {e6.max():6.2f}; scan transformed p50/max {np.percentile(e7, 50):6.2f} {e7.max():6.2f}")
    for it in range(8):
        b = r[it][r[it, :, 0] > 0]
        if len(b) == 0 or b[:, 0].min() < t0:
            continue
        st, out, gat, upd = (b[:, 0] - t0) * 0.01, (b[:, 1] - t0) * 0.01, (b[:, 2] - t0) * 0.01, (b[:, 3] - t0) * 0.01
        print(f"  pass {it}: {len(b)} blocks; start min/max {st.min():6.2f} {st.max():6.2f}; block partials out p50/max {np.percentile(out, 50):6.2f} {out.max():6.2f}; "
              f"all gathered min/max {gat.min():6.2f} {gat.max():6.2f}; update done min/max {upd.min():6.2f} {upd.max():6.2f}")
e = w[40:48].astype(np.int64)
if e[7] > 0:
    print(f"ekf_step_wave cycles (mean over {e[7]} updates): w + gauss-jordan {e[0] / e[7]:.0f}  y + solution {e[2] / e[7]:.0f}  exp / state {e[3] / e[7]:.0f}  (prior [-] iterate: overlapped with the gather)")
