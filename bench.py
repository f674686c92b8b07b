#!/usr/bin/env python3
"""Benchmark of the PQ linear hot path on MI355X.

Workload (BASELINE.json configs[1]): the PQ-quantised linear layers of Llama-3-8B -- qkv_proj,
o_proj, gate_up_proj, down_proj of all 32 decoder layers, K=256, sub-dim 8, one codebook per output
partition, bf16 activations -- called through PQLinearMethod.apply() behind the tensor-parallel linear
layers.  One step = one pass of a batch of `--tokens` synthetic tokens through those 128 apply() calls
(every layer has its own weights: 832 MiB of codes, more than the 256 MiB Infinity Cache, so codes
stream from HBM).  With --gpus N the layers are sharded column/row-parallel over N ranks (TP=N) and
every row-parallel output is summed with an RCCL all-reduce; total work is fixed ("strong" scaling).

Prints ONE JSON line on rank 0.  `value` = effective TFLOP/s of the whole job,
2*T*sum(N*Kin) / step time; `hbm_gbps` and `tokens_per_s` are the same measurement in the metric's
other two units.  `roofline` is for the dominant kernel (gate_up_proj), timed with HIP events on the
launch stream; `cpu_baseline` times the torch-CPU restatement of the reference's CPU algorithm
(oracle/torch_restatement.py) on a bounded sample on this box's host cores; `decode` repeats the
step measurement at a small token count (HBM-bound regime).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec
MFMA_PEAK_TFLOPS = 2500.0     # dense bf16/fp16 MFMA spec

LLAMA3_8B = dict(hidden=4096, inter=14336, heads=32, kv_heads=8, head_dim=128, layers=32)
K_ENTRIES, SUB_DIM, NUM_CODEBOOKS = 256, 8, 1


def algorithmic_bytes(T, N, Kin, P, K=K_ENTRIES, d=SUB_DIM, M=NUM_CODEBOOKS, bias=False):
    """Bytes one apply() must move: codes + codebooks + scales + x + y (+ bias)."""
    cb = 1 if K <= 256 else 2
    e = 2
    return N * (Kin // d) * M * cb + P * M * K * d * e + N * e + T * Kin * e + T * N * e + (N * e if bias else 0)


def build_stack(tp_size, device, dtype, cfg=LLAMA3_8B, layers=None):
    """The four PQ linears of every decoder layer, through the plugin surface."""
    from vllm_pq_amd import PQConfig
    from vllm_pq_amd.host.linear import MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear
    qc = PQConfig(SUB_DIM, K_ENTRIES.bit_length() - 1, NUM_CODEBOOKS, 1)
    g = torch.Generator(device="cpu").manual_seed(1234)
    stack = []
    for _ in range(layers if layers is not None else cfg["layers"]):
        block = {
            "qkv_proj": QKVParallelLinear(cfg["hidden"], cfg["head_dim"], cfg["heads"], cfg["kv_heads"], bias=False,
                                          params_dtype=dtype, quant_config=qc),
            "o_proj": RowParallelLinear(cfg["heads"] * cfg["head_dim"], cfg["hidden"], bias=False,
                                        params_dtype=dtype, quant_config=qc),
            "gate_up_proj": MergedColumnParallelLinear(cfg["hidden"], [cfg["inter"]] * 2, bias=False,
                                                       params_dtype=dtype, quant_config=qc),
            "down_proj": RowParallelLinear(cfg["inter"], cfg["hidden"], bias=False, params_dtype=dtype,
                                           quant_config=qc),
        }
        for layer in block.values():
            layer.codes.data = torch.randint(-128, 128, layer.codes.shape, generator=g, dtype=torch.int8)
            layer.codebooks.data = torch.randn(layer.codebooks.shape, generator=g).to(dtype)
            layer.scales.data = (torch.rand(layer.scales.shape, generator=g) + 0.5).to(dtype)
            layer.to(device)
            layer.quant_method.process_weights_after_loading(layer)
        stack.append(block)
    return stack


def shapes_of(block):
    """(name, N_rank, Kin_rank, P) per linear of one decoder layer on this rank."""
    out = []
    for name, layer in block.items():
        N, G, _ = layer.codes.shape
        out.append((name, N, G * SUB_DIM, len(layer.codebooks.output_partition_sizes)))
    return out


def make_inputs(block, T, device, dtype):
    g = torch.Generator(device="cpu").manual_seed(4321)
    return {name: torch.randn(T, layer.codes.shape[1] * SUB_DIM, generator=g).to(dtype).to(device)
            for name, layer in block.items()}


# ---- launch log (profiling aid) -------------------------------------------------------------------------
# With --launch-log every PQ GEMM launch of the process is recorded in launch order as (layer, T, N, Kin, kernel);
# a graph replay appends what its capture recorded.  tools/proftrace.py aligns the list with the GEMM dispatches of a
# rocprofv3 kernel trace of the same run, so durations are keyed by problem rather than by (kernel, grid) -- the
# persistent tiled kernel launches 256 blocks whatever the layer.
LAUNCH_LOG = None


def enable_launch_log(stack, dtype):
    """Wraps the operator layer's GEMM entry points (every PQ GEMM launch of the process goes through one of
    them, whichever plugin method issued it), so the log holds exactly one record per kernel dispatch."""
    global LAUNCH_LOG
    from vllm_pq_amd import ops
    LAUNCH_LOG = []
    names = {}
    for name, layer in stack[0].items():
        names[(layer.codes.shape[0], layer.codes.shape[1])] = name

    def record(x, codes, codebooks, parts, packed, fused=None):
        N, G, M = codes.shape
        T = x.numel() // x.shape[-1]
        kern = ops.gemm_kernel_name(T, N, G, M, codebooks.shape[1], codebooks.shape[3], codes.element_size(),
                                    dtype, list(parts) or [N], packed=packed) + (fused or "")
        LAUNCH_LOG.append([names.get((N, G), "?"), T, N, G * codebooks.shape[3], kern])

    plain, packed, fused = ops.pq_gemm, ops.pq_gemm_packed, ops.pq_gemm_packed_silu_mul

    def gemm(input, codes, codebooks, scales, codebook_partition_sizes, bias=None):
        record(input, codes, codebooks, codebook_partition_sizes, False)
        return plain(input, codes, codebooks, scales, codebook_partition_sizes, bias)

    def gemm_packed(input, packed_codes, codes, codebooks, scales, codebook_partition_sizes, bias=None):
        record(input, codes, codebooks, codebook_partition_sizes, True)
        return packed(input, packed_codes, codes, codebooks, scales, codebook_partition_sizes, bias)

    def gemm_packed_silu_mul(input, packed_codes, codes, codebooks, scales, bias=None):
        record(input, codes, codebooks, [codes.shape[0] // 2] * 2, True, fused="+silu_mul")
        return fused(input, packed_codes, codes, codebooks, scales, bias)

    prepared = ops.pq_gemm_packed_prepared

    def gemm_packed_prepared(desc, input, bias=None):
        record(input, desc.codes, desc.codebooks, [desc.parts[i] for i in range(desc.nparts)], True)
        return prepared(desc, input, bias)

    ops.pq_gemm_packed_prepared = gemm_packed_prepared
    add_norm, normed = ops.pq_gemm_packed_add_norm, ops.pq_gemm_packed_normed

    def gemm_packed_add_norm(input, packed_codes, codes, codebooks, scales, codebook_partition_sizes, bias, residual,
                             norm_weight):
        record(input, codes, codebooks, codebook_partition_sizes, True, fused="+add_norm")
        return add_norm(input, packed_codes, codes, codebooks, scales, codebook_partition_sizes, bias, residual, norm_weight)

    def gemm_packed_normed(xw, ssq, eps, packed_codes, codes, codebooks, scales, codebook_partition_sizes, bias):
        record(xw, codes, codebooks, codebook_partition_sizes, True, fused="+normed")
        return normed(xw, ssq, eps, packed_codes, codes, codebooks, scales, codebook_partition_sizes, bias)

    ops.pq_gemm, ops.pq_gemm_packed, ops.pq_gemm_packed_silu_mul = gemm, gemm_packed, gemm_packed_silu_mul
    ops.pq_gemm_packed_add_norm, ops.pq_gemm_packed_normed = gemm_packed_add_norm, gemm_packed_normed
    normed_act = ops.pq_gemm_packed_normed_silu_mul

    def gemm_packed_normed_silu_mul(xw, ssq, eps, packed_codes, codes, codebooks, scales, bias):
        record(xw, codes, codebooks, [codes.shape[0] // 2] * 2, True, fused="+normed+silu_mul")
        return normed_act(xw, ssq, eps, packed_codes, codes, codebooks, scales, bias)

    ops.pq_gemm_packed_normed_silu_mul = gemm_packed_normed_silu_mul


class Graph:
    """torch.cuda.CUDAGraph whose replays also replay the launch log of its capture."""

    def __init__(self, fn, **capture_args):
        self.g = torch.cuda.CUDAGraph()
        n0 = len(LAUNCH_LOG) if LAUNCH_LOG is not None else 0
        # Capture by hand rather than with torch.cuda.graph(): when the captured function fails (a collective that
        # cannot be captured), that context manager's exit raises from capture_end and leaves the capturing side stream
        # current, so that the next GPU call of the process fails too.  Here the stream context always unwinds.
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.g.capture_begin(**capture_args)
            try:
                self.out = fn()
            except BaseException:
                try:
                    self.g.capture_end()
                except Exception:
                    pass
                raise
            self.g.capture_end()
        torch.cuda.current_stream().wait_stream(side)
        self.logged = LAUNCH_LOG[n0:] if LAUNCH_LOG is not None else []
        if LAUNCH_LOG is not None:
            del LAUNCH_LOG[n0:]            # capture launches nothing

    def replay(self):
        self.g.replay()
        if LAUNCH_LOG is not None:
            LAUNCH_LOG.extend(self.logged)


def run_step(stack, inputs):
    out = None
    for block in stack:
        for name, layer in block.items():
            out, _ = layer(inputs[name])
    return out


class StepRunner:
    """Runs `fn` eagerly or, when capture works, as one HIP graph replay (as vLLM does for decode).  With several
    ranks the graph holds the RCCL all-reduces too; capture runs in thread-local error mode (the process group's
    watchdog thread polls events meanwhile), and the ranks agree afterwards: if capture failed on any of them, all
    of them launch eagerly, in this same process."""

    def __init__(self, fn, use_graph, dist_on=False):
        self.fn, self.graph = fn, None
        if use_graph:
            try:
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    fn()
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                self.graph = Graph(fn, capture_error_mode="thread_local")
            except Exception as e:  # capture unsupported (e.g. a collective): fall back to eager launches
                print(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
                self.graph = None
                torch.cuda.synchronize()
            if dist_on:
                import torch.distributed as dist
                ok = torch.tensor([1 if self.graph is not None else 0], dtype=torch.int32,
                                  device="cpu" if dist.get_backend() == "gloo" else "cuda")
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 0:
                    self.graph = None

    def __call__(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self.fn()


def timed_steps(runner, steps, warmup, dist_on):
    import torch.distributed as dist
    for _ in range(warmup):
        runner()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        runner()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def time_dominant_kernel(stack, inputs, name, reps):
    """Average duration of one launch of the dominant linear's kernel: `reps` back-to-back launches over
    the layers' distinct weights between two HIP events on the launch stream (no collectives); the median
    of five such measurements."""
    layers = [b[name] for b in stack]
    x = inputs[name]
    qm = layers[0].quant_method
    for l in layers[:4]:
        qm.apply(l, x)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    def launches():
        for i in range(reps):
            qm.apply(layers[i % len(layers)], x)
    g = Graph(launches)              # graph replay: back-to-back launches without host gaps
    g.replay()
    torch.cuda.synchronize()
    per_launch = []
    for _ in range(5):               # median of five replays: one replay of short launches is noisy
        e0.record(stream)
        g.replay()
        e1.record(stream)
        torch.cuda.synchronize()
        per_launch.append(e0.elapsed_time(e1) * 1e-3 / reps)
    return sorted(per_launch)[len(per_launch) // 2]


def host_cpu():
    """Model string, vector ISA levels and topology of the host CPU (what the torch-CPU operators can use): sockets,
    physical cores and logical CPUs of the machine from /proc/cpuinfo, and the logical CPUs this process may run on."""
    model, flags = "unknown", set()
    phys, logical, sockets = set(), 0, set()
    try:
        with open("/proc/cpuinfo") as f:
            pid = cid = None
            for line in f:
                if line.startswith("model name") and model == "unknown":
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("flags") and not flags:
                    flags = set(line.split(":", 1)[1].split())
                elif line.startswith("processor"):
                    logical += 1
                    pid = cid = None
                elif line.startswith("physical id"):
                    pid = line.split(":", 1)[1].strip()
                    sockets.add(pid)
                elif line.startswith("core id"):
                    cid = line.split(":", 1)[1].strip()
                    phys.add((pid, cid))
    except OSError:
        pass
    isa = [k for k in ("avx2", "avx512f", "avx512_bf16", "avx512_fp16", "amx_bf16", "amx_tile") if k in flags]
    try:
        allowed = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = None
    topo = {"sockets": len(sockets) or None, "physical_cores": len(phys) or None, "logical_cpus": logical or None,
            "cpus_allowed": allowed}
    return model, isa, topo


def cpu_baseline(T, dtype, budget_s=12.0, min_calls=5):
    """The reference's CPU algorithm (torch-CPU restatement) on one decoder layer's four linears at T tokens: every
    linear is called at least `min_calls` times (and until its share of about `budget_s` seconds is used), the
    rate is the layer's FLOPs over the sum of the four MEDIAN call times."""
    from oracle.torch_restatement import pq_linear_cpu
    cfg = LLAMA3_8B
    q, kv = cfg["heads"] * cfg["head_dim"], cfg["kv_heads"] * cfg["head_dim"]
    lin = [("qkv_proj", [q, kv, kv], cfg["hidden"]), ("o_proj", [cfg["hidden"]], q),
           ("gate_up_proj", [cfg["inter"]] * 2, cfg["hidden"]), ("down_proj", [cfg["hidden"]], cfg["inter"])]
    g = torch.Generator().manual_seed(99)
    cores = torch.get_num_threads()
    flops, med_secs, calls, spent = 0.0, 0.0, 0, 0.0
    for i, (name, parts, kin) in enumerate(lin):
        n = sum(parts)
        codes = torch.randint(-128, 128, (n, kin // SUB_DIM, 1), generator=g, dtype=torch.int8)
        cb = torch.randn(len(parts), K_ENTRIES, 1, SUB_DIM, generator=g).to(dtype)
        sc = (torch.rand(n, 1, 1, 1, generator=g) + 0.5).to(dtype)
        x = torch.randn(T, kin, generator=g).to(dtype)
        if i == 0:
            pq_linear_cpu(x[:8], codes[:256], cb[:1], sc[:256], [256], None)      # warm the operators up
        share = budget_s * (n * kin) / sum(sum(p_) * k_ for _, p_, k_ in lin)
        times = []
        while len(times) < min_calls or (sum(times) < share and len(times) < 200):
            t0 = time.perf_counter()
            pq_linear_cpu(x, codes, cb, sc, parts, None)
            times.append(time.perf_counter() - t0)
        times.sort()
        med_secs += times[len(times) // 2]
        spent += sum(times)
        calls += len(times)
        flops += 2.0 * T * n * kin
    model, isa, topo = host_cpu()
    return {"value": flops / med_secs / 1e12, "unit": "TFLOP/s", "cores": cores, "kind": "port",
            "cpu_model": model, "cpu_isa": isa, "cpu_topology": topo, "tokens": T, "layer_ms": med_secs * 1e3,
            "sample": f"{calls} apply() calls over qkv/o/gate_up/down of one Llama-3-8B decoder layer at T={T} "
                      f"(at least {min_calls} per linear, median call time of each; {spent:.1f} s of CPU work, "
                      f"{cores} threads), torch-CPU restatement of the reference's generic_dequantize_gemm"}


def time_neighbours(T, device, dtype, world, cfg=LLAMA3_8B, reps=5):
    """The element-wise kernels either side of the linears (silu_and_mul after gate_up_proj, fused_add_rms_norm
    before qkv_proj / gate_up_proj) at this token count: achieved HBM GB/s on their algorithmic bytes, timed
    with events around a HIP-graph replay of back-to-back launches over rotating buffers."""
    from vllm_pq_amd import ops
    inter, hidden = cfg["inter"] // world, cfg["hidden"]
    out = {}

    def timed(fn, sets, nbytes):
        fn(sets[0])
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for s_ in sets:
                fn(s_)
        g.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e-3 / len(sets))
        return {"launch_us": best * 1e6, "algorithmic_bytes": nbytes, "achieved": nbytes / best / 1e9,
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": nbytes / best / 1e9 / HBM_PEAK_GBPS}

    n_sets = max(2, min(8, int(600e6 / max(1, T * inter * 6)) + 1))
    sets = [(torch.randn(T, 2 * inter, device=device).to(dtype), torch.empty(T, inter, device=device, dtype=dtype))
            for _ in range(n_sets)]
    out["silu_and_mul"] = timed(lambda s_: ops.silu_and_mul(s_[1], s_[0]), sets, T * inter * 2 * 3)
    del sets
    n_sets = max(2, min(16, int(600e6 / max(1, T * hidden * 8)) + 1))
    w = torch.ones(hidden, device=device, dtype=dtype)
    sets = [(torch.randn(T, hidden, device=device).to(dtype), torch.randn(T, hidden, device=device).to(dtype))
            for _ in range(n_sets)]
    out["fused_add_rms_norm"] = timed(lambda s_: ops.fused_add_rms_norm(s_[0], s_[1], w, 1e-5), sets,
                                      T * hidden * 2 * 4 + hidden * 2)
    return out


def layer_chain(stack, T, device, dtype, cfg=LLAMA3_8B, reps=5):
    """The linears with their element-wise neighbours as a decoder layer chains them (models/llama.py:89-93,
    :272-284): add + norm -> qkv_proj -> o_proj (the query heads stand in for the attention output) -> add + norm ->
    gate_up_proj with SiLU-and-mul -> down_proj, for every layer, one HIP-graph replay; best of `reps`."""
    from vllm_pq_amd.host.layers import RMSNorm
    hidden = cfg["hidden"]
    norms = []
    for _ in stack:
        pair = (RMSNorm(hidden, 1e-5).to(device), RMSNorm(hidden, 1e-5).to(device))
        for n in pair:
            n.weight.data = n.weight.data.to(dtype)
        norms.append(pair)
    x0 = torch.randn(T, hidden, device=device).to(dtype)
    q = cfg["heads"] * cfg["head_dim"]

    def step():
        h, res = x0.clone(), x0.clone()
        for blk, (n1, n2) in zip(stack, norms):
            h, res = n1(h, res)
            qkv, _ = blk["qkv_proj"](h)
            h, _ = blk["o_proj"](qkv[:, :q])
            h, res = n2(h, res)
            gu = blk["gate_up_proj"]
            a = gu.quant_method.apply_silu_and_mul(gu, h)
            h, _ = blk["down_proj"](a)
        return h

    # The same chain with every add + norm carried by the linears either side of it (PQLinearMethod.apply_add_norm /
    # apply_normed: decode batches, one rank): o_proj -> [add, norm] -> gate_up_proj, down_proj -> [add, norm] -> the
    # next layer's qkv_proj; the first layer's input norm and the activation stay launches of their own.
    qm = stack[0]["o_proj"].quant_method
    fusable = all(qm.norm_pair_preferred(b["o_proj"], b["gate_up_proj"], T) and
                  qm.norm_pair_preferred(b["down_proj"], b["qkv_proj"], T) for b in stack[:1])

    def step_fused():
        h, res = x0.clone(), x0.clone()
        h, res = norms[0][0](h, res)
        qkv, _ = stack[0]["qkv_proj"](h)
        for i, (blk, (n1, n2)) in enumerate(zip(stack, norms)):
            xw, ssq = qm.apply_add_norm(blk["o_proj"], qkv[:, :q], res, n2.weight.data)
            a = qm.apply_normed_silu_and_mul(blk["gate_up_proj"], xw, ssq, 1e-5)
            if i + 1 < len(stack):
                xw, ssq = qm.apply_add_norm(blk["down_proj"], a, res, norms[i + 1][0].weight.data)
                qkv = qm.apply_normed(stack[i + 1]["qkv_proj"], xw, ssq, 1e-5)
            else:
                h, _ = blk["down_proj"](a)
        return h

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        g = Graph(fn)
        g.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    separate = timed(step)
    if fusable:
        fused = timed(step_fused)
        return {"ms": fused, "ms_separate_norms": separate, "layers": len(stack), "launch": "hipgraph",
                "steps_per_layer": ["o_proj+add (+ sum of squares)", "gate_up_proj (x 1/rms) + silu_and_mul",
                                    "down_proj+add (+ sum of squares)", "next qkv_proj (x 1/rms)"],
                "steps_per_layer_separate": ["fused_add_rms_norm", "qkv_proj", "o_proj", "fused_add_rms_norm",
                                             "gate_up_proj+silu_and_mul", "down_proj"]}
    best = separate
    return {"ms": best, "layers": len(stack), "steps_per_layer": ["fused_add_rms_norm", "qkv_proj", "o_proj",
                                                                 "fused_add_rms_norm", "gate_up_proj+silu_and_mul",
                                                                 "down_proj"], "launch": "hipgraph"}


def mid_batches(stack, device, dtype, tokens=(128, 256), reps=32):
    """Mid-size batches (continuous-batching decode, chunked prefill): every linear of a decoder layer at these token
    counts, one launch timed as the dominant kernel is (HIP events around a graph replay of back-to-back launches over
    the layers' distinct weights), its fraction of the binding roofline, and the four-linear sum."""
    from vllm_pq_amd import ops
    out = {}
    for T in tokens:
        inputs = make_inputs(stack[0], T, device, dtype)
        rec, total_us, total_peak_us = {}, 0.0, 0.0
        for (name, n, kin, p) in shapes_of(stack[0]):
            t = time_dominant_kernel(stack, inputs, name, max(reps, len(stack)))
            flops, byts = 2.0 * T * n * kin, algorithmic_bytes(T, n, kin, p)
            peak_s = max(flops / (MFMA_PEAK_TFLOPS * 1e12), byts / (HBM_PEAK_GBPS * 1e9))
            parts = stack[0][name].codebooks.output_partition_sizes
            rec[name] = {"launch_us": t * 1e6, "frac": peak_s / t,
                         "bound": "mfma" if flops / (MFMA_PEAK_TFLOPS * 1e12) >= byts / (HBM_PEAK_GBPS * 1e9) else "hbm",
                         "kernel": ops.gemm_kernel_name(T, n, kin // SUB_DIM, NUM_CODEBOOKS, K_ENTRIES, SUB_DIM, 1, dtype, parts,
                                                        packed=getattr(stack[0][name], "pq_packed_codes", None) is not None)}
            total_us += t * 1e6
            total_peak_us += peak_s * 1e6
        rec["four_linears_us"] = total_us
        rec["four_linears_frac"] = total_peak_us / total_us
        out[f"T{T}"] = rec
    return out


def mfma_ceiling(device, dtype, seconds=1.5):
    """What a bare MFMA loop reaches on THIS device with operands drawn like the benchmark's (normal values rounded
    to the act dtype, in registers, no memory traffic; tools/pq_probe.hip): the chip lowers its clock under MFMA load
    by an amount that depends on the operands' bit patterns, so the nominal peak is not attainable on such data.
    None when the probe library was not built."""
    import ctypes
    path = os.path.join(ROOT, "tools", "libpq_probe.so")
    if not os.path.exists(path) or dtype != torch.bfloat16:
        return None
    lib = ctypes.CDLL(path)
    lib.pq_probe_mfma.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.pq_probe_mfma.restype = ctypes.c_int
    n_in, blocks, iters = 65536, 256, 4000
    ops_in = torch.randn(n_in * 8, device=device).to(dtype)
    out = torch.empty(blocks * 512, device=device, dtype=torch.float32)
    stream = torch.cuda.current_stream().cuda_stream

    def launch():
        rc = lib.pq_probe_mfma(ops_in.data_ptr(), n_in, out.data_ptr(), blocks, iters, stream)
        assert rc == 0, f"pq_probe_mfma failed ({rc})"
    flops = blocks * 8 * iters * 32 * 16384.0
    launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:      # clocks settle under sustained load
        for _ in range(20):
            launch()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 40
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    return flops * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12


def library_gemm(layer, T, device, dtype, k_flops, reps=20):
    """The dense problem the fused kernel replaces, run by the vendor's library in the same process on the same device:
    y = x @ W^T with W the layer's dequantised weight in the act dtype (torch.matmul -> hipBLASLt / rocBLAS), and the
    dequantisation that route needs first (the reference's path above 6 tokens, aqlm.py:141-145).  A yardstick for the
    clock the chip holds on this kind of data, not part of the product path."""
    from vllm_pq_amd import ops
    parts = list(layer.codebooks.output_partition_sizes)
    w = ops.pq_dequant(layer.codes, layer.codebooks, parts)
    x = torch.randn(T, w.shape[1], device=device).to(dtype)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e-3 / reps
            best = t if best is None else min(best, t)
        return best
    t_mm = timed(lambda: torch.matmul(x, w.t()))
    t_dq = timed(lambda: ops.pq_dequant(layer.codes, layer.codebooks, parts))
    return {"gemm_us": t_mm * 1e6, "gemm_tflops": k_flops / t_mm / 1e12, "dequant_us": t_dq * 1e6,
            "dequant_plus_gemm_us": (t_mm + t_dq) * 1e6,
            "what": "torch.matmul (hipBLASLt / rocBLAS) on the layer's dequantised weight, same shapes, same run; "
                    "pq_dequant of the layer: the two steps of the reference's route at this batch size"}


def profiled_traffic(key):
    """(HBM bytes per launch of a kernel, where the number comes from): measured by the committed rocprofv3 counter
    passes (profiles/traffic_r*.json, newest round first) -- not in this run, PMC counters need the profiler --;
    (None, None) when that launch was not profiled."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json")), reverse=True):
        try:
            with open(path) as f:
                doc = json.load(f)
            rec = doc.get(key)
            if rec:
                return rec["hbm_bytes"], {"file": os.path.relpath(path, ROOT), "commit": doc.get("commit", "unrecorded"),
                                          "box": doc.get("box", "unrecorded"),
                                          "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh "
                                                 "(committed; not measured in this run)"}
        except (OSError, ValueError, KeyError):
            pass
    return None, None


def split_times(stack, inputs, T, steps, warmup, device, dtype, dist_on, use_graph):
    """Several ranks: where a step's time goes.  `kernel_ms` = the step with the row-parallel layers' all-reduces
    switched off (the 128 GEMM launches alone), `allreduce_ms` = the step's 64 all-reduces alone, on [T, hidden]
    tensors; each timed like the step itself (barrier + synchronize either side, max over ranks)."""
    import torch.distributed as dist
    rows = [l for b in stack for l in b.values() if getattr(l, "reduce_results", False)]
    for l in rows:
        l.reduce_results = False
    try:
        k = timed_steps(StepRunner(lambda: run_step(stack, inputs), use_graph, dist_on), steps, warmup, dist_on) / steps
    finally:
        for l in rows:
            l.reduce_results = True
    hidden = rows[0].output_size
    bufs = [torch.zeros(T, hidden, device=device, dtype=dtype) for _ in range(4)]    # sums of zeros stay finite

    from vllm_pq_amd.host import parallel_state

    def reduces():      # what a row-parallel layer issues behind its GEMM
        for i in range(len(rows)):
            if parallel_state.get_tensor_model_parallel_world_size() > 1:
                parallel_state.tensor_model_parallel_all_reduce(bufs[i % 4], in_place=True)
            elif dist_on:   # one-rank rehearsal of the distributed path: the collective itself
                dist.all_reduce(bufs[i % 4], op=dist.ReduceOp.SUM)
    a = timed_steps(StepRunner(reduces, use_graph, dist_on), steps, warmup, dist_on) / steps
    return {"kernel_ms": k * 1e3, "allreduce_ms": a * 1e3, "kernel_plus_allreduce_ms": (k + a) * 1e3,
            "allreduces_per_step": len(rows), "allreduce_bytes": T * hidden * 2,
            "rccl_ranks": dist.get_world_size() if dist_on else 1,
            "small_allreduce": "xgmi-peer-buffers" if parallel_state._XGMI_AR is not None else "rccl"}


def measure(stack, T, steps, warmup, device, dtype, world, dist_on, use_graph, split=False):
    inputs = make_inputs(stack[0], T, device, dtype)
    if dist_on:         # the ranks start a measurement together (rank 0 alone ran the single-rank probes before)
        import torch.distributed as dist
        torch.cuda.synchronize()
        dist.barrier()
    runner = StepRunner(lambda: run_step(stack, inputs), use_graph, dist_on)
    dt = timed_steps(runner, steps, warmup, dist_on)
    per_step = dt / steps
    # whole-job work: the unsharded model's FLOPs; bytes: what all ranks must move
    flops = sum(2.0 * T * n * kin for (_, n, kin, _) in shapes_of(stack[0])) * len(stack) * world
    byts = sum(algorithmic_bytes(T, n, kin, p) for (_, n, kin, p) in shapes_of(stack[0])) * len(stack) * world
    return dict(per_step=per_step, tflops=flops / per_step / 1e12, gbps=byts / per_step / 1e9,
                tokens_per_s=T / per_step, launch="hipgraph" if runner.graph is not None else "eager",
                inputs=inputs,
                split=split_times(stack, inputs, T, steps, warmup, device, dtype, dist_on, use_graph) if world > 1 or split else None)


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes of this script
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment, a free local port), wait for
    them, pass their output through and return non-zero if any of them failed.  The parent has made no GPU call
    (importing torch makes none) and makes none: it only waits."""
    import socket
    import subprocess
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in env:
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(("127.0.0.1", 0))
            env["MASTER_PORT"] = str(s.getsockname()[1])
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["WORLD_SIZE"] = env["LOCAL_WORLD_SIZE"] = str(n)
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        # rank 0's stdout is the job's (the JSON line); the others' goes to stderr so the line stays alone
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        # a rank that dies leaves the others waiting in a collective: stop them once any has failed
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0:
                    print(f"[bench] rank {r} exited with status {code}", file=sys.stderr)
                    rc = rc or (code if code > 0 else 1)
            if rc:
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:       # only after a failure (or an interrupt): exactly the children started above
                p.terminate()
                rc = rc or 1
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tokens", type=int, default=4096, help="tokens per step (batch x seq)")
    ap.add_argument("--decode-tokens", type=int, default=16, help="token count of the secondary decode measurement")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--layers", type=int, default=None, help="decoder layers (default: all 32)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--eager-collectives", action="store_true",
                    help="with --gpus > 1, launch the steps eagerly; by default they are captured, RCCL all-reduces "
                         "included, into a HIP graph (falling back to eager launches when capture fails)")
    ap.add_argument("--xgmi-allreduce", action="store_true",
                    help="with --gpus > 1, sum messages of up to 2 MiB with the library's peer-buffer all-reduce "
                         "instead of RCCL (not yet measured on multi-GPU hardware: off by default)")
    ap.add_argument("--split-times", action="store_true",
                    help="report kernel_ms / allreduce_ms on one GPU too (always reported with --gpus > 1)")
    ap.add_argument("--launch-log", default=None,
                    help="write the process's PQ GEMM launches, in launch order, to this JSON file (for tools/proftrace.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--no-neighbours", action="store_true", help="skip the element-wise kernels' measurement")
    ap.add_argument("--no-mid", action="store_true", help="skip the mid-size batches (128 and 256 tokens per linear)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="with --gpus > 1 on a box with ONE GPU: every rank uses device 0 and the collectives go through "
                         "gloo -- the multi-rank code path end to end, its numbers meaningless (the line says so)")
    ap.add_argument("--dry-ranks", action="store_true",
                    help="every rank prints its RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as one JSON line and exits "
                         "without touching a GPU (exercises the rank launcher)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process never touches a GPU, it starts the N ranks and waits
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if os.environ.get("PQ_BENCH_TRACE_AFTER"):      # diagnosis of a hang: every thread's stack after so many seconds, then exit
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["PQ_BENCH_TRACE_AFTER"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or "RANK" in os.environ
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_ranks:
        print(json.dumps({"rank": rank, "local_rank": local_rank, "world": world,
                          "master": f"{os.environ.get('MASTER_ADDR', '')}:{os.environ.get('MASTER_PORT', '')}"}),
              flush=True)
        return
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no ROCm device is visible")
    dev_index = 0 if args.rehearse_one_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}[args.dtype]

    import torch.distributed as dist
    from vllm_pq_amd.host import parallel_state
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if args.rehearse_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        parallel_state.initialize_model_parallel(world)
        if args.xgmi_allreduce and world > 1:
            parallel_state.enable_xgmi_all_reduce()

    t_start = time.perf_counter()

    def note(what):     # progress on stderr: a run that stops shows where (stdout carries the JSON line only)
        print(f"[bench] rank {rank}: {what} ({time.perf_counter() - t_start:.1f} s)", file=sys.stderr, flush=True)

    from vllm_pq_amd import ops
    stack = build_stack(world, device, dtype, layers=args.layers)
    note("stack built")
    T = args.tokens
    if args.launch_log:
        enable_launch_log(stack, dtype)
    # a step is one HIP graph replay, collectives included: 128 GEMM launches and 64 all-reduces issued from Python
    # would make a decode step host-bound
    use_graph = not args.no_graph and (world == 1 or not args.eager_collectives)
    if args.rehearse_one_gpu and world > 1:
        use_graph = False       # a gloo collective cannot be captured, and one that fails half-way inside a capture leaves the ranks out of step

    main_m = measure(stack, T, args.steps, args.warmup, device, dtype, world, dist_on, use_graph, args.split_times)
    note(f"{T}-token steps timed")

    # dominant kernel: gate_up_proj (largest share of FLOPs and of code bytes)
    dom = "gate_up_proj"
    _, n_d, kin_d, p_d = [s for s in shapes_of(stack[0]) if s[0] == dom][0]
    reps = max(len(stack), 32)
    t_k = time_dominant_kernel(stack, main_m["inputs"], dom, reps)
    k_flops = 2.0 * T * n_d * kin_d
    k_bytes = algorithmic_bytes(T, n_d, kin_d, p_d)
    dom_parts = stack[0][dom].codebooks.output_partition_sizes
    dom_packed = getattr(stack[0][dom], "pq_packed_codes", None) is not None
    kname = ops.gemm_kernel_name(T, n_d, kin_d // SUB_DIM, NUM_CODEBOOKS, K_ENTRIES, SUB_DIM, 1, dtype, dom_parts,
                                 packed=dom_packed)
    # binding roofline of that launch: whichever of HBM and MFMA takes longer at peak
    if k_flops / (MFMA_PEAK_TFLOPS * 1e12) >= k_bytes / (HBM_PEAK_GBPS * 1e9):
        roof = {"bound": "mfma", "achieved": k_flops / t_k / 1e12, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s"}
    else:
        roof = {"bound": "hbm", "achieved": k_bytes / t_k / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s"}
    roof["frac"] = roof["achieved"] / roof["peak"]
    roof["traffic"], roof["traffic_source"] = profiled_traffic(f"{dom}@T{T}:{kname}") if world == 1 else (None, None)
    roof.update({"kernel": f"{kname}:{dom}", "launch_us": t_k * 1e6, "launches_timed": reps,
                 "algorithmic_bytes": k_bytes, "algorithmic_flops": k_flops})
    if roof["bound"] == "mfma" and rank == 0:
        ceil = mfma_ceiling(device, dtype)
        if ceil:
            # measured in this run on this device: a bare MFMA loop on operands drawn like the benchmark's
            roof["measured_ceiling"] = {"value": ceil, "unit": "TFLOP/s", "frac_of_ceiling": roof["achieved"] / ceil,
                                        "what": "bare v_mfma_f32_16x16x32_bf16 loop, normal operands in registers, "
                                                "2 waves per SIMD, after 1.5 s of load"}

    if roof["bound"] == "mfma" and rank == 0 and world == 1:
        lg = library_gemm(stack[0][dom], T, device, dtype, k_flops)
        lg["fused_over_library"] = lg["gemm_us"] / roof["launch_us"]
        lg["fused_over_dequant_plus_library"] = lg["dequant_plus_gemm_us"] / roof["launch_us"]
        roof["library_gemm"] = lg

    note("dominant launch timed")
    decode = None
    if not args.no_decode and args.decode_tokens != T:
        Td = args.decode_tokens
        dm = measure(stack, Td, max(args.steps, 20), args.warmup, device, dtype, world, dist_on, use_graph, args.split_times)
        t_kd = time_dominant_kernel(stack, dm["inputs"], dom, reps)
        kb = algorithmic_bytes(Td, n_d, kin_d, p_d)
        decode = {"tokens": Td, "ms_per_step": dm["per_step"] * 1e3, "tflops": dm["tflops"], "hbm_gbps": dm["gbps"],
                  "tokens_per_s": dm["tokens_per_s"], "launch": dm["launch"],
                  **(dm["split"] or {}),
                  "roofline": {"bound": "hbm", "achieved": kb / t_kd / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": kb / t_kd / 1e9 / HBM_PEAK_GBPS,
                               "traffic": profiled_traffic(f"{dom}@T{Td}:" + ops.gemm_kernel_name(
                                   Td, n_d, kin_d // SUB_DIM, 1, K_ENTRIES, SUB_DIM, 1, dtype, dom_parts,
                                   packed=dom_packed))[0] if world == 1 else None,
                               "traffic_source": profiled_traffic(f"{dom}@T{Td}:" + ops.gemm_kernel_name(
                                   Td, n_d, kin_d // SUB_DIM, 1, K_ENTRIES, SUB_DIM, 1, dtype, dom_parts,
                                   packed=dom_packed))[1] if world == 1 else None,
                               "kernel": ops.gemm_kernel_name(Td, n_d, kin_d // SUB_DIM, 1, K_ENTRIES, SUB_DIM, 1, dtype,
                                                              dom_parts, packed=dom_packed) + ":" + dom,
                               "launch_us": t_kd * 1e6}}

    note("decode steps timed")
    neighbours = None
    if not args.no_neighbours:
        neighbours = {f"T{T}": time_neighbours(T, device, dtype, world)}
        if not args.no_decode and args.decode_tokens != T:
            neighbours[f"T{args.decode_tokens}"] = time_neighbours(args.decode_tokens, device, dtype, world)

    chain = None
    if world == 1 and not args.no_neighbours:
        chain = {f"T{T}": layer_chain(stack, T, device, dtype)}
        if not args.no_decode and args.decode_tokens != T:
            chain[f"T{args.decode_tokens}"] = layer_chain(stack, args.decode_tokens, device, dtype)

    mid = None
    if world == 1 and not args.no_mid:
        mid = mid_batches(stack, device, dtype)

    note("neighbours, chain and mid-size batches timed")
    # the peer-buffer all-reduce's status word after everything timed: a wait that gave up poisons its output and stays on record
    ar_status = None
    if world > 1 and parallel_state._XGMI_AR is not None:
        torch.cuda.synchronize()
        try:
            parallel_state._XGMI_AR.check()
            ar_status = "ok"
        except RuntimeError as e:
            ar_status = f"FAILED: {e}"

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(T, dtype)
        if decode is not None:      # the decode regime: the same algorithm re-dequantises the weights for 16 tokens
            decode["cpu_baseline"] = cpu_baseline(args.decode_tokens, dtype, budget_s=8.0)

    if rank == 0:
        line = {
            "metric": "PQ dequant+GEMM effective TFLOP/s (Llama-3-8B PQ linear stack)",
            "value": main_m["tflops"], "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": main_m["per_step"] * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "Llama-3-8B PQ linears (qkv/o/gate_up/down x %d layers), K=256, sub-dim=8, M=1"
                                   % len(stack),
                       "tokens_per_step": T, "parallelism": f"tp{world}", "launch": main_m["launch"],
                       "apply_calls_per_step": 4 * len(stack),
                       "tokens_per_s_covers": "the PQ linear layers only (no attention, norms, sampling): tokens of a "
                                              "step / step time of the %d apply() calls" % (4 * len(stack))},
            "hbm_gbps": main_m["gbps"], "tokens_per_s": main_m["tokens_per_s"],
            "roofline": roof, "cpu_baseline": cpu, "decode": decode, "neighbours": neighbours,
            "layer_chain": chain, "mid_batches": mid,
        }
        if ar_status is not None:
            line["pq_ar_status"] = ar_status
        if args.rehearse_one_gpu:
            line["rehearsal"] = "all ranks on one GPU, collectives through gloo: code path only, the numbers mean nothing"
        if main_m["split"]:
            line.update(main_m["split"])
        print(json.dumps(line), flush=True)
        if args.launch_log:
            with open(args.launch_log, "w") as f:
                json.dump(LAUNCH_LOG, f)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
