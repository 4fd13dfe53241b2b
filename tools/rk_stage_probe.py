"""Which stage of the throughput-mode eval path moves ranks?  The trained TVR-dims model of tools/rk_gate_tvr.py,
scored from raw features by: parity (fp32-grade towers), k4_only (bf16 input projection, fp32-grade towers), fast (K4 + fused bf16
tower K5 from fp32 h0), resident with fp32 h0 rows, resident with bf16 h0 rows (eval_epoch's default).  Per mode and seed: mean /
max score error against the fp32 CPU oracle, NET recall deltas and GROSS crossings per cut.

    python tools/rk_stage_probe.py [--steps 600] [--seeds 3] [--nv 4096] [--nq 8192] [--out gpurun_out/rk_stage_probe.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (os.path.join(ROOT, "dl-dkd_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--nv", type=int, default=4096)
    ap.add_argument("--nq", type=int, default=8192)
    ap.add_argument("--sigma", type=float, default=6.0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "rk_stage_probe.json"))
    a = ap.parse_args()
    import rk_gate
    import rk_gate_tvr as G
    from dldkd_amd import ops
    P, Pt = G.maps()
    m, losses = G.train_model(a.steps, a.sigma, P, Pt, log=print)
    res = {"train_steps": a.steps, "n_videos": a.nv, "n_queries": a.nq, "seeds": []}
    variants = (("parity", "parity", None), ("k4_only", "k4_only", None), ("fast", "fast", None), ("resident_fp32_h0", "resident", False),
                ("resident_bf16_h0", "resident", True))
    for s in range(a.seeds):
        d = {k: v.cpu() for k, v in G.make_pairs(500 + s, a.nv, a.nq // a.nv, 64, 8, a.sigma, P, Pt, dev="cuda:0").items()}
        ref, _, _ = rk_gate.oracle_scores(m, d, threads=32, chunk=512)
        rk_ref, r_ref = rk_gate.recalls(ref, d["gt"])
        row = {"seed": 500 + s, "oracle": rk_ref}
        for name, mode, h16 in variants:
            old = ops.RESIDENT_H0_H16
            if h16 is not None:
                ops.RESIDENT_H0_H16 = h16
            try:
                fused, _, _ = rk_gate.hip_scores(m, d, mode)
            finally:
                ops.RESIDENT_H0_H16 = old
            rk, r = rk_gate.recalls(fused, d["gt"])
            row[name] = {"net_queries": [int(round((x - y) * a.nq / 100.0)) for x, y in zip(rk, rk_ref)],
                         "gross_queries": [int(((r <= k) != (r_ref <= k)).sum()) for k in (1, 5, 10, 100)],
                         "mean_abs_score_err": float((fused - ref).abs().mean()), "max_abs_score_err": float((fused - ref).abs().max())}
            print(f"seed {500 + s} {name:18s} mean err {row[name]['mean_abs_score_err']:.3e}  net {row[name]['net_queries']}  "
                  f"gross {row[name]['gross_queries']}", flush=True)
        res["seeds"].append(row)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
