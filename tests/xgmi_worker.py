"""One rank of the peer-buffer all-reduce test (tests/test_xgmi_all_reduce.py): started as a fresh process, two
of them share cuda:0.  Usage: python tests/xgmi_worker.py RANK WORLD PORT OUTDIR"""
import json
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def inputs(world, shape, dtype, seed):
    import torch
    out = []
    for r in range(world):
        g = torch.Generator().manual_seed(seed * 100 + r)
        out.append(torch.randn(shape, generator=g).to(dtype))
    return out


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    res = {"rank": rank, "ok": False, "checks": []}
    try:
        import torch
        import torch.distributed as dist
        from vllm_pq_amd.host import parallel_state as ps
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        ps.initialize_model_parallel(world)
        assert ps.enable_xgmi_all_reduce()
        ar = ps._XGMI_AR
        dev = torch.device("cuda", 0)

        # known answer, as the reference's comm-op test (tests/distributed/test_comm_ops.py:30-38)
        t = (torch.arange(8, dtype=torch.float32) * (rank + 1)).to(dev)
        assert ar.should_use(t)
        got = ps.tensor_model_parallel_all_reduce(t)
        torch.cuda.synchronize()
        want = torch.arange(8, dtype=torch.float32) * sum(r + 1 for r in range(world))
        assert torch.equal(got.cpu(), want), got
        assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32) * (rank + 1))     # out of place
        res["checks"].append("known answer")

        # one shot (128 KiB) and two shots (1 MiB, 2 MiB), every dtype, out of place and in place, many calls in a
        # row with changing sizes (the lanes' halves alternate per call)
        cases = [((16, 4096), torch.bfloat16), ((128, 4096), torch.float16), ((7, 4096), torch.bfloat16),
                 ((256, 4096), torch.bfloat16), ((1, 8), torch.float16), ((64, 4096), torch.float32),
                 ((33, 1000), torch.float32)]
        for it in range(3):
            for ci, (shape, dt) in enumerate(cases):
                xs = inputs(world, shape, dt, 10 * it + ci)
                acc = xs[0].float()
                for x in xs[1:]:
                    acc = acc + x.float()
                want = acc.to(dt)
                mine = xs[rank].to(dev)
                assert ar.should_use(mine), (shape, dt)
                if (it + ci) % 2:
                    got = ps.tensor_model_parallel_all_reduce(mine, in_place=True)
                    assert got is mine
                else:
                    got = ps.tensor_model_parallel_all_reduce(mine)
                torch.cuda.synchronize()
                assert torch.equal(got.cpu(), want), f"{shape} {dt} call {it}: {int((got.cpu() != want).sum())} differ"
        ar.check()
        res["checks"].append("sizes and dtypes")

        # not taken: too large, or not a multiple of 16 bytes
        assert not ar.should_use(torch.empty(3 << 20, dtype=torch.bfloat16, device=dev))
        assert not ar.should_use(torch.empty(7, dtype=torch.bfloat16, device=dev))

        # captured into a HIP graph and replayed: the call counters live on the device
        x = inputs(world, (16, 4096), torch.bfloat16, 77)
        buf = x[rank].to(dev)
        out = torch.empty_like(buf)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            ar.all_reduce(buf, out)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        dist.barrier()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            ar.all_reduce(buf, out)
            ar.all_reduce(out, out)
        want1 = (x[0].float() + x[1].float()).to(torch.bfloat16) if world == 2 else None
        for _ in range(3):
            g.replay()
            torch.cuda.synchronize()
            if want1 is not None:
                want2 = (want1.float() * world).to(torch.bfloat16)
                assert torch.equal(out.cpu(), want2)
        ar.check()
        res["checks"].append("graph replay")
        # the sum with the decoder block's residual add + RMSNorm in its epilogue: bit for bit the all-reduce followed
        # by the library's fused_add_rms_norm (itself pinned to the reference's forward_native by tests/test_layers.py)
        from vllm_pq_amd import ops
        for ci, (T, H, dt) in enumerate([(16, 4096, torch.bfloat16), (1, 4096, torch.float16), (5, 8192, torch.bfloat16),
                                         (33, 1024, torch.bfloat16), (64, 4096, torch.bfloat16), (3, 16384, torch.float16),
                                         (7, 1000, torch.bfloat16)]):
            xs = inputs(world, (T, H), dt, 500 + ci)
            g = torch.Generator().manual_seed(900 + ci)
            res0 = torch.randn(T, H, generator=g).to(dt).to(dev)
            w = (torch.rand(H, generator=g) + 0.5).to(dt).to(dev)
            mine = xs[rank].to(dev)
            assert ar.add_norm_supported(mine, res0), (T, H, dt)
            res_f = res0.clone()
            keep = mine.clone()
            xw = ps.tensor_model_parallel_all_reduce_add_norm(mine, res_f, w, 1e-5)
            torch.cuda.synchronize()
            assert torch.equal(mine, keep)                      # the fused call leaves its input alone
            summed = ps.tensor_model_parallel_all_reduce(mine)
            res_u = res0.clone()
            ops.fused_add_rms_norm(summed, res_u, w, 1e-5)
            torch.cuda.synchronize()
            assert torch.equal(res_f.view(torch.int16), res_u.view(torch.int16)), f"residual {T}x{H} {dt}"
            bad = (xw.view(torch.int16) != summed.view(torch.int16)).nonzero()
            assert bad.numel() == 0, (f"normed {T}x{H} {dt}: {bad.shape[0]} differ, first at {bad[0].tolist()}: "
                                      f"{xw[tuple(bad[0].tolist())].item()} vs {summed[tuple(bad[0].tolist())].item()}; "
                                      f"rows {sorted(set(bad[:, 0].tolist()))[:8]} cols {bad[:8, 1].tolist()}")
        # the same through the row-parallel layer: o_proj -> sum over ranks + residual add + RMSNorm in one call, against
        # forward() followed by the norm kernel
        from vllm_pq_amd import PQConfig
        from vllm_pq_amd.host.linear import RowParallelLinear
        torch.manual_seed(4242)
        row = RowParallelLinear(1024 * world, 2048, bias=True, params_dtype=torch.bfloat16, quant_config=PQConfig(8, 8, 1, 1))
        row.codes.data = torch.randint(-128, 128, row.codes.shape, dtype=torch.int8)
        row.codebooks.data = torch.randn(row.codebooks.shape).to(torch.bfloat16)
        row.scales.data = (torch.rand(row.scales.shape) + 0.5).to(torch.bfloat16)
        row.bias.data = torch.randn(row.bias.shape).to(torch.bfloat16)
        row.to(dev)
        row.quant_method.process_weights_after_loading(row)
        for T in (1, 16, 40):
            g = torch.Generator().manual_seed(77 + T + rank)
            xr = torch.randn(T, 1024, generator=g).to(torch.bfloat16).to(dev)
            g2 = torch.Generator().manual_seed(7)
            res0 = torch.randn(T, 2048, generator=g2).to(torch.bfloat16).to(dev)
            w = (torch.rand(2048, generator=g2) + 0.5).to(torch.bfloat16).to(dev)
            res_f = res0.clone()
            xw = row.forward_add_norm(xr, res_f, w, 1e-5)
            y, _ = row(xr)
            res_u = res0.clone()
            ops.fused_add_rms_norm(y, res_u, w, 1e-5)
            torch.cuda.synchronize()
            assert torch.equal(res_f.view(torch.int16), res_u.view(torch.int16)), f"row layer residual T={T}"
            assert torch.equal(xw.view(torch.int16), y.view(torch.int16)), f"row layer normed T={T}"
        res["checks"].append("row-parallel layer with fused sum + add + norm")

        # not taken (too large for one shot / rows not a multiple of 8): the two steps run, same interface
        big = torch.randn(256, 4096).to(torch.bfloat16).to(dev)
        assert not ar.add_norm_supported(big, torch.empty_like(big))
        rb = torch.zeros_like(big)
        out = ps.tensor_model_parallel_all_reduce_add_norm(big.clone(), rb, torch.ones(4096, dtype=torch.bfloat16, device=dev), 1e-5)
        torch.cuda.synchronize()
        assert out.shape == big.shape and torch.isfinite(out.float()).all()
        ar.check()
        res["checks"].append("fused add + norm")

        # LAST (it ruins the communicator): a peer that never arrives.  Rank 0 alone issues a call with a short
        # timeout: its output must be NaN everywhere, never a stale sum, and the status must say so.
        dist.barrier()
        if rank == 0:
            ar.set_timeout_ms(200)
            lone = torch.ones(16, 4096, dtype=torch.bfloat16, device=dev)
            got = ar.all_reduce(lone)
            r2 = torch.zeros(4, 4096, dtype=torch.bfloat16, device=dev)
            xw2 = ar.all_reduce_add_norm(lone[:4].contiguous(), r2, torch.ones(4096, dtype=torch.bfloat16, device=dev), 1e-5)
            torch.cuda.synchronize()
            assert torch.isnan(got.float()).all(), "a timed-out all-reduce must not return a sum"
            assert torch.isnan(xw2.float()).all() and torch.isnan(r2.float()).all()
            try:
                ar.check()
                raise AssertionError("check() did not report the timeout")
            except RuntimeError as e:
                assert "timed out" in str(e)
        dist.barrier()
        try:
            ps.destroy_model_parallel()             # close() reports the timeout once more on the rank that saw it
            assert rank != 0, "close() did not report the timeout"
        except RuntimeError as e:
            assert rank == 0 and "timed out" in str(e)
        res["checks"].append("timeout poisons the output")
        dist.destroy_process_group()
        res["ok"] = True
    except BaseException as e:  # noqa: BLE001 - the parent reports it
        res["error"] = f"{type(e).__name__}: {e}\n{traceback.format_exc()}"
    with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    main()
