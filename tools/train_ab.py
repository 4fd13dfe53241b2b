"""Does the bf16 training mode TRAIN like the parity mode?  Same initial weights, same batches, same triplet draws:
N steps of the TVR-dims planted-pairs task (tools/rk_gate_tvr.py) in parity mode (fp32-grade GEMMs: losses within 1e-4 of the
reference) - twice, because the step's fp32 atomics make two parity runs differ: their distance is the noise floor - and once in
bf16 mode (the 2.4-ms step).  Every model is then evaluated the same way: fp32 oracle towers + oracle scoring on the CPU, on eval
sets it never saw.  Reported: R@1/5/10/100 of each model per eval seed, bf16 - parity and parity_b - parity deltas, the loss
curves' windowed means and their relative distance.

    python tools/train_ab.py [--steps 1500] [--seeds 3] [--nv 4096] [--nq 8192] [--out profiles/r05/train_ab.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (os.path.join(ROOT, "dl-dkd_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)


def windowed(losses, win):
    """[(step, loss)] sampled every k steps -> means over windows of `win` samples."""
    v = [l for _, l in losses]
    return [sum(v[i:i + win]) / len(v[i:i + win]) for i in range(0, len(v) - win + 1, win)]


def run(steps=1500, seeds=3, nv=4096, nq=8192, sigma=6.0, every=10, win=10, chunk=512, log=print, runs=("parity", "parity_b", "bf16")):
    import rk_gate
    import rk_gate_tvr as G
    P, Pt = G.maps()
    models, curves, secs = {}, {}, {}
    for name in runs:
        t0 = time.time()
        m, losses = G.train_model(steps, sigma, P, Pt, seed=0, precision={"bf16": "bf16", "mixed": "mixed"}.get(name, "fp32"), every=every, log=None)
        models[name], curves[name], secs[name] = m, losses, round(time.time() - t0, 2)
        log(f"trained {name}: {steps} steps in {secs[name]} s, loss {losses[0][1]:.3f} -> {losses[-1][1]:.3f}")
    res = {"task": "TVR dims (Dv 3072, Dq 768), planted pairs, 128 videos x 5 captions per step, dropout 0.1, hard negatives, soft labels",
           "steps": steps, "train_seconds": secs, "eval": {"n_videos": nv, "n_queries": nq, "oracle": "fp32 CPU restatement, towers + scoring"},
           "loss_window_means": {k: windowed(v, win) for k, v in curves.items()}, "seeds": []}
    base = res["loss_window_means"]["parity"]
    for k in runs[1:]:
        res[f"loss_curve_max_rel_diff_{k}_vs_parity"] = max(abs(a - b) / max(abs(b), 1e-9) for a, b in zip(res["loss_window_means"][k], base))
    import torch
    for s in range(seeds):
        d = {k: v.cpu() for k, v in G.make_pairs(500 + s, nv, nq // nv, 64, 8, sigma, P, Pt, dev="cuda:0").items()}
        row = {"seed": 500 + s}
        for name in runs:
            ref, _, _ = rk_gate.oracle_scores(models[name], d, threads=32, chunk=chunk)
            row[name] = rk_gate.recalls(ref, d["gt"])[0]
        for k in runs[1:]:
            row[f"{k}_minus_parity"] = [a - b for a, b in zip(row[k], row["parity"])]
        res["seeds"].append(row)
        log(f"eval seed {500 + s}: " + "  ".join(f"{k} {['%.2f' % x for x in row[k]]}" for k in runs) + "  " +
            "  ".join(f"{k}-parity {['%+.2f' % x for x in row[k + '_minus_parity']]}" for k in runs[1:]))
        del d
    for k in runs[1:]:
        res[f"worst_abs_recall_delta_{k}_vs_parity"] = max(abs(x) for r in res["seeds"] for x in r[f"{k}_minus_parity"])
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--nv", type=int, default=4096)
    ap.add_argument("--nq", type=int, default=8192)
    ap.add_argument("--out", default="")
    ap.add_argument("--runs", default="parity,parity_b,bf16", help="comma-separated: parity first, then any of parity_b, bf16, mixed")
    a = ap.parse_args()
    res = run(a.steps, a.seeds, a.nv, a.nq, runs=tuple(a.runs.split(",")))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k not in ("seeds", "loss_window_means")}))
