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
        ps.destroy_model_parallel()
        dist.destroy_process_group()
        res["ok"] = True
    except BaseException as e:  # noqa: BLE001 - the parent reports it
        res["error"] = f"{type(e).__name__}: {e}\n{traceback.format_exc()}"
    with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    main()
