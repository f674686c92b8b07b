"""Offline product quantiser: dense linear weights -> the `codes` / `codebooks` / `scales` tensors the PQ
linear method loads, and a checkpoint writer for them.

Format (the AQLM tensor layout the reference's loaders use, vllm/model_executor/layers/quantization/aqlm.py:253-316):
  codes      [N, G, M]      int8 (K <= 256) or int16, signed storage of the unsigned index (index = value mod K,
                            aqlm.py:32-34), G = in_features / d
  codebooks  [M, K, 1, d]   one set per unfused projection; the model's fused layers stack them (aqlm.py:279-296)
  scales     [N, 1, 1, 1]   one per output row
  W[n, g*d:(g+1)*d] ~= scales[n] * sum_m codebooks[m, index(n, g, m), 0, :]          (aqlm.py:37-74)

Algorithm: rows are normalised by their RMS, the d-dimensional sub-vectors of all rows are clustered with
k-means (k-means++ seeding on a sample, Lloyd iterations, empty clusters re-seeded on the worst-fitted
points); further codebooks (M > 1) are fitted to the residual of the previous ones (residual quantisation);
the row scale is then refitted by least squares against the reconstruction, and codebooks/codes and scales
are alternated a few more times (`refits`) from the centroids already found.  Everything runs in torch on the
device of the weight, so a GPU is used when the weight is on one; there is no dependence on the kernels.

This is tooling beside the hot path: the reference ships no quantiser (its AQLM checkpoints come from a
separate project), so there is no reference code this restates.
"""
import json
import os
import re
from typing import Dict, Iterable, List, Optional, Tuple

import torch

from .pq import get_int_dtype

__all__ = ["pq_quantize_weight", "pq_reconstruct", "quantize_state_dict", "save_pq_checkpoint", "quantization_config"]


def _assign(x: torch.Tensor, centroids: torch.Tensor, chunk: int = 1 << 16) -> Tuple[torch.Tensor, torch.Tensor]:
    """Nearest centroid of every row of x [n, d]: (index [n], squared distance [n])."""
    idx = torch.empty(x.shape[0], dtype=torch.long, device=x.device)
    dist = torch.empty(x.shape[0], dtype=x.dtype, device=x.device)
    c2 = (centroids * centroids).sum(1)
    for s in range(0, x.shape[0], chunk):
        xs = x[s:s + chunk]
        # |x - c|^2 = |x|^2 - 2 x.c + |c|^2 ; |x|^2 does not change the argmin
        score = c2.unsqueeze(0) - 2.0 * (xs @ centroids.t())
        m, i = score.min(1)
        idx[s:s + chunk] = i
        dist[s:s + chunk] = m + (xs * xs).sum(1)
    return idx, dist.clamp_min_(0)


def _kmeans(x: torch.Tensor, k: int, iters: int, gen: torch.Generator,
            init: Optional[torch.Tensor] = None) -> torch.Tensor:
    """k centroids [k, d] for the rows of x [n, d]; `init` continues from given centroids (no seeding)."""
    n, d = x.shape
    if init is not None:
        return _lloyd(x, init.clone(), iters)
    if n <= k:      # fewer points than centroids: every point is its own centroid, the rest repeat
        reps = (k + n - 1) // n
        return x.repeat(reps, 1)[:k].clone()
    # k-means++ seeding on a sample (the full D^2 sweep is k passes over the data)
    sample = x[torch.randperm(n, generator=gen, device="cpu")[:min(n, 32 * k)].to(x.device)]
    cent = torch.empty(k, d, dtype=x.dtype, device=x.device)
    cent[0] = sample[torch.randint(sample.shape[0], (1,), generator=gen).item()]
    d2 = ((sample - cent[0]) ** 2).sum(1)
    for j in range(1, k):
        p = d2 / d2.sum().clamp_min(1e-30)
        pick = torch.multinomial(p.cpu(), 1, generator=gen).item() if float(d2.sum()) > 0 else \
            torch.randint(sample.shape[0], (1,), generator=gen).item()
        cent[j] = sample[pick]
        d2 = torch.minimum(d2, ((sample - cent[j]) ** 2).sum(1))
    return _lloyd(x, cent, iters)


def _lloyd(x: torch.Tensor, cent: torch.Tensor, iters: int) -> torch.Tensor:
    k = cent.shape[0]
    for _ in range(iters):
        idx, dist = _assign(x, cent)
        sums = torch.zeros_like(cent).index_add_(0, idx, x)
        cnt = torch.zeros(k, dtype=x.dtype, device=x.device).index_add_(0, idx, torch.ones_like(dist))
        empty = cnt == 0
        cent = torch.where(empty.unsqueeze(1), cent, sums / cnt.clamp_min(1).unsqueeze(1))
        ne = int(empty.sum())
        if ne and x.shape[0] > k:      # re-seed empty clusters on the worst-fitted points
            worst = dist.topk(ne).indices
            cent[empty] = x[worst]
    return cent


def pq_quantize_weight(weight: torch.Tensor, d: int = 8, nbits: int = 8, num_codebooks: int = 1, iters: int = 25,
                       seed: int = 0, dtype: Optional[torch.dtype] = None, refits: int = 2
                       ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Quantise one projection `weight` [out_features, in_features] -> (codes, codebooks, scales)."""
    if weight.dim() != 2:
        raise ValueError(f"weight must be [out_features, in_features], got {tuple(weight.shape)}")
    N, Kin = weight.shape
    if Kin % d:
        raise ValueError(f"in_features={Kin} is not a multiple of the sub-vector length d={d}")
    if not 1 <= nbits <= 16:
        raise ValueError("nbits must be in 1..16")
    out_dtype = dtype or (weight.dtype if weight.dtype in (torch.float16, torch.bfloat16) else torch.bfloat16)
    K, G = 1 << nbits, Kin // d
    gen = torch.Generator().manual_seed(seed)
    w = weight.detach().to(torch.float32)
    scale = w.pow(2).mean(1).sqrt()
    books: List[torch.Tensor] = []
    for rnd in range(refits + 1):
        # alternate: codebooks and codes for the rows divided by their scale, then the scale for those codes
        safe = torch.where(scale.abs() > 1e-12, scale, torch.ones_like(scale))
        resid = (w / safe.unsqueeze(1)).reshape(N * G, d)
        idx_all, prev, books = [], books, []
        for m in range(num_codebooks):
            cent = _kmeans(resid, K, iters if rnd == 0 else max(1, iters // 4), gen, prev[m] if prev else None)
            cent = cent.to(out_dtype).to(torch.float32)  # the kernels see the rounded entries: fit against those
            idx, _ = _assign(resid, cent)
            resid = resid - cent[idx]
            idx_all.append(idx.view(N, G))
            books.append(cent)
        recon = torch.zeros(N, Kin, dtype=torch.float32, device=w.device)
        for idx, cent in zip(idx_all, books):
            recon += cent[idx].reshape(N, Kin)
        # least-squares row scale against the (unscaled) reconstruction
        scale = (w * recon).sum(1) / (recon * recon).sum(1).clamp_min(1e-30)
    codes = torch.stack(idx_all, dim=2).to(get_int_dtype(nbits))          # wraps: index 200 -> int8 -56
    codebooks = torch.stack(books, 0).view(num_codebooks, K, 1, d).to(out_dtype)
    scales = scale.view(N, 1, 1, 1).to(out_dtype)
    return codes.contiguous(), codebooks.contiguous(), scales.contiguous()


def pq_reconstruct(codes: torch.Tensor, codebooks: torch.Tensor, scales: torch.Tensor) -> torch.Tensor:
    """fp32 dense weight of one projection from its PQ tensors (plain torch; for error reports and tests)."""
    N, G, M = codes.shape
    K, d = codebooks.shape[1], codebooks.shape[3]
    idx = codes.to(torch.int64) % K
    w = torch.zeros(N, G, d, dtype=torch.float32, device=codes.device)
    for m in range(M):
        w += codebooks[m, :, 0, :].to(torch.float32)[idx[:, :, m]]
    return w.reshape(N, G * d) * scales.view(N, 1).to(torch.float32)


def quantization_config(d: int, nbits: int, num_codebooks: int = 1) -> Dict[str, int]:
    """The `quantization_config` entry of the HF config.json that selects this method (PQConfig.from_config)."""
    return {"quant_method": "pq", "in_group_size": d, "nbits_per_codebook": nbits,
            "num_codebooks": num_codebooks, "out_group_size": 1}


def quantize_state_dict(state: Dict[str, torch.Tensor], include: str = r".*(q_proj|k_proj|v_proj|o_proj|gate_proj|"
                        r"up_proj|down_proj|out_proj|fc1|fc2)\.weight$", d: int = 8, nbits: int = 8,
                        num_codebooks: int = 1, iters: int = 25, seed: int = 0, device: Optional[str] = None,
                        report: Optional[List[Tuple[str, float]]] = None) -> Dict[str, torch.Tensor]:
    """Replace every `<module>.weight` whose name matches `include` by `<module>.codes/.codebooks/.scales`
    (the checkpoint names the reference's model loaders expect, vllm/model_executor/models/llama.py:371-433);
    other tensors pass through.  `report` collects (name, relative Frobenius error)."""
    pat = re.compile(include)
    out: Dict[str, torch.Tensor] = {}
    for i, (name, t) in enumerate(state.items()):
        if t.dim() == 2 and pat.match(name) and t.shape[1] % d == 0:
            w = t.to(device) if device else t
            codes, cb, sc = pq_quantize_weight(w, d, nbits, num_codebooks, iters, seed + i)
            if report is not None:
                err = (pq_reconstruct(codes, cb, sc) - w.float()).norm() / w.float().norm().clamp_min(1e-30)
                report.append((name, float(err)))
            base = name[:-len(".weight")]
            out[base + ".codes"], out[base + ".codebooks"], out[base + ".scales"] = codes.cpu(), cb.cpu(), sc.cpu()
        else:
            out[name] = t
    return out


def save_pq_checkpoint(state: Dict[str, torch.Tensor], out_dir: str, d: int, nbits: int, num_codebooks: int = 1,
                       base_config: Optional[dict] = None) -> None:
    """Write `model.safetensors` and a `config.json` carrying the quantization_config into `out_dir`."""
    from safetensors.torch import save_file
    os.makedirs(out_dir, exist_ok=True)
    save_file({k: v.contiguous() for k, v in state.items()}, os.path.join(out_dir, "model.safetensors"))
    cfg = dict(base_config or {})
    cfg["quantization_config"] = quantization_config(d, nbits, num_codebooks)
    with open(os.path.join(out_dir, "config.json"), "w") as f:
        json.dump(cfg, f, indent=1)


def _main(argv: Optional[Iterable[str]] = None) -> int:
    import argparse
    from safetensors.torch import load_file
    ap = argparse.ArgumentParser(description="Product-quantise the linear weights of a safetensors checkpoint.")
    ap.add_argument("input", help="directory with *.safetensors (+ config.json) or one .safetensors file")
    ap.add_argument("output", help="directory to write")
    ap.add_argument("--sub-dim", type=int, default=8)
    ap.add_argument("--nbits", type=int, default=8)
    ap.add_argument("--num-codebooks", type=int, default=1)
    ap.add_argument("--iters", type=int, default=25)
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    args = ap.parse_args(argv)
    files = [args.input] if args.input.endswith(".safetensors") else \
        sorted(os.path.join(args.input, f) for f in os.listdir(args.input) if f.endswith(".safetensors"))
    state: Dict[str, torch.Tensor] = {}
    for f in files:
        state.update(load_file(f))
    base_cfg = None
    cfg_path = os.path.join(args.input if os.path.isdir(args.input) else os.path.dirname(args.input), "config.json")
    if os.path.exists(cfg_path):
        with open(cfg_path) as f:
            base_cfg = json.load(f)
    report: List[Tuple[str, float]] = []
    q = quantize_state_dict(state, d=args.sub_dim, nbits=args.nbits, num_codebooks=args.num_codebooks,
                            iters=args.iters, device=args.device, report=report)
    for name, err in report:
        print(f"{name}: relative error {err:.4f}")
    save_pq_checkpoint(q, args.output, args.sub_dim, args.nbits, args.num_codebooks, base_cfg)
    return 0


if __name__ == "__main__":
    raise SystemExit(_main())
