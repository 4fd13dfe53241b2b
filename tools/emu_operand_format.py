"""Emulation (plain torch fp32 on the GPU, no HIP kernels): how far do the eval-path scores move when every GEMM operand of the
throughput-mode towers is rounded to bf16 versus to fp16 (IEEE half: 10 mantissa bits instead of 7, range 6e-5 .. 65504)?  The towers
are the oracle's formulas (oracle/dldkd_oracle.py) with a rounding r() applied to both operands of every product and to the stored
h0 rows; the scorer's operands are rounded to bf16 in every variant (K1 stays a bf16 kernel).  Prints mean |score error| against the
unrounded fp32 scores and the largest magnitude every rounded operand class takes (fp16 range check).

    python tools/emu_operand_format.py [--steps 600]
"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (os.path.join(ROOT, "dl-dkd_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)
import torch
import torch.nn.functional as F

MAXABS = {}


def rounder(kind):
    def r(t, tag):
        MAXABS[tag] = max(MAXABS.get(tag, 0.0), float(t.abs().max()))
        if kind == "fp32":
            return t
        return t.to(torch.bfloat16 if kind == "bf16" else torch.float16).float()
    return r


def ln(x, w, b):
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


def tower(p, feat, mask, proj, enc, pos, r, round_h0):
    h = ln(feat, p[proj + ".LayerNorm.weight"], p[proj + ".LayerNorm.bias"])
    h = torch.relu(r(h, "inproj_x") @ r(p[proj + ".net.1.weight"], "inproj_w").t() + p[proj + ".net.1.bias"])
    if round_h0:
        h = r(h, "h0")
    L = h.shape[1]
    h1 = ln(h + p[pos + ".position_embeddings.weight"][:L].unsqueeze(0), p[pos + ".LayerNorm.weight"], p[pos + ".LayerNorm.bias"])
    N, _, D = h1.shape
    dh = D // 4
    x = r(h1, "h1")
    sp = lambda t: t.view(N, L, 4, dh).permute(0, 2, 1, 3)   # noqa: E731
    q = sp(x @ r(p[enc + ".self.query.weight"], "w").t() + p[enc + ".self.query.bias"])
    k = sp(x @ r(p[enc + ".self.key.weight"], "w").t() + p[enc + ".self.key.bias"])
    v = sp(x @ r(p[enc + ".self.value.weight"], "w").t() + p[enc + ".self.value.bias"])
    s = (r(q, "q") @ r(k, "k").transpose(-1, -2)) / math.sqrt(dh) + ((1 - mask) * -10000.0).view(N, 1, 1, L)
    a = torch.softmax(s, -1)
    ctx = (r(a, "p") @ r(v, "v")).permute(0, 2, 1, 3).reshape(N, L, D)
    h2 = r(ctx, "ctx") @ r(p[enc + ".output.dense.weight"], "w").t() + p[enc + ".output.dense.bias"]
    return ln(h2 + x, p[enc + ".output.LayerNorm.weight"], p[enc + ".output.LayerNorm.bias"])


def scores(p, d, kind, round_h0, dev):
    r = rounder(kind)
    out = []
    for pre in ("", "exp_"):
        gs = []
        for s in range(0, d["vid"].shape[0], 512):
            v, m = d["vid"][s:s + 512].to(dev), d["vmask"][s:s + 512].to(dev)
            h = tower(p, v, m, pre + "visual_input_proj", pre + "visual_encoder", pre + "visual_pos_embed", r, round_h0)
            gs.append(r(h, "h2") @ r(p[pre + "out_mapping_linear.weight"], "w").t() + p[pre + "out_mapping_linear.bias"])
        g = torch.cat(gs)
        qs = []
        for s in range(0, d["words"].shape[0], 2048):
            w, m = d["words"][s:s + 2048].to(dev), d["qmask"][s:s + 2048].to(dev)
            h = tower(p, w, m, pre + "query_input_proj", pre + "query_encoder", pre + "query_pos_embed", r, False)
            lg = (h @ p[pre + "modular_vector_mapping.weight"].t()).squeeze(-1)
            lg = lg * m + (1 - m) * -1e10
            qs.append(torch.einsum("nl,nld->nd", torch.softmax(lg, 1), h))
        q = torch.cat(qs)
        gn, qn = F.normalize(g, dim=-1), F.normalize(q, dim=-1)
        if kind != "fp32":                                  # the scorer (K1) takes bf16 rows in every variant
            gn, qn = gn.to(torch.bfloat16).float(), qn.to(torch.bfloat16).float()
        vm = d["vmask"].to(dev)
        sc = torch.empty(q.shape[0], g.shape[0], device=dev)
        for s in range(0, q.shape[0], 1024):
            cs = torch.einsum("md,nld->mnl", qn[s:s + 1024], gn)
            sc[s:s + 1024] = (cs * vm + (1 - vm) * -1e10).max(-1).values
        out.append(sc)
    return 0.7 * out[0] + 0.3 * out[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--nv", type=int, default=2048)
    ap.add_argument("--nq", type=int, default=4096)
    a = ap.parse_args()
    import rk_gate
    import rk_gate_tvr as G
    dev = "cuda:0"
    P, Pt = G.maps()
    m, _ = G.train_model(a.steps, 6.0, P, Pt, log=None)
    p = {k: v.detach().float() for k, v in m.state_dict().items()}
    d = {k: v.cpu() for k, v in G.make_pairs(500, a.nv, a.nq // a.nv, 64, 8, 6.0, P, Pt, dev=dev).items()}
    with torch.no_grad():
        ref = scores(p, d, "fp32", False, dev)
        rk_ref, r_ref = rk_gate.recalls(ref.cpu(), d["gt"])
        print("fp32 reference recalls", ["%.2f" % x for x in rk_ref])
        for kind, h0 in (("bf16", False), ("bf16", True), ("fp16", False), ("fp16", True)):
            MAXABS.clear()
            sc = scores(p, d, kind, h0, dev)
            rk, r = rk_gate.recalls(sc.cpu(), d["gt"])
            gross = [int(((r <= k) != (r_ref <= k)).sum()) for k in (1, 5, 10, 100)]
            print(f"{kind} operands, h0 rows {'16-bit' if h0 else 'fp32'}: mean |score err| {float((sc - ref).abs().mean()):.3e}  max {float((sc - ref).abs().max()):.3e}  "
                  f"gross crossings {gross}")
        print("largest |value| per rounded operand class:", {k: round(v, 4) for k, v in MAXABS.items()})
        tiny = {k: float((t.abs() < 6.1e-5).float().mean()) for k, t in (("inproj_w", p["visual_input_proj.net.1.weight"]), ("x", d["vid"][:64]))}
        print("fraction of elements below fp16's smallest normal (6.1e-5):", tiny)


if __name__ == "__main__":
    main()
