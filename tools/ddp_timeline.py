"""World-1 RCCL timeline of one training_all iteration (TrainStep.step_all) - VERDICT r04 item 2: is the gradient exchange
launched DURING the last domain's backward?

    run      python3 tools/ddp_timeline.py run
             one process = rank 0 of a 1-rank NCCL (= RCCL) group with FPLX_DDP_FORCE=1: every collective of the data-parallel
             path runs (full-batch loss sums, bucketed all-reduce, BatchNorm segments), the benchmark network and shape; prints
             HIP-event times (stream order) of every collective's launch against the end of each domain's backward
    analyze  python3 tools/ddp_timeline.py analyze <kernel_trace.csv>
             per iteration (delimited by the optimiser launch): start of the first / last RCCL kernel against the end of the
             last backward kernel of the iteration (the stem's weight gradient), and the RCCL kernels that START before it
"""
import csv
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def run():
    os.environ.setdefault("FPLX_DDP_FORCE", "1")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("LOCAL_RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    sys.path.insert(0, os.path.join(ROOT, "fpl-plus_amd"))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import fplx
    import bench
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    net = fplx.UNet2D5_dsbn(dict(bench.NET)).to(dev)
    net._ensure_flat()
    ts = fplx.TrainStep(net, (1.0, 0.0, 0.0, 0.0), True, lr=1e-4, weight_decay=1e-5)
    assert ts.reducer.enabled and ts.overlap_all
    batches = []
    for d in (0, 1):
        x, lab = bench.synth_batch(bench.SHAPE, d, dev)
        batches.append({"image": x, "label_prob": lab})
    for _ in range(4):
        ts.step_all(batches)
    torch.cuda.synchronize()
    # A 1-rank RCCL all-reduce is a no-op on the device (no kernel in a trace), so the evidence is taken with HIP events in stream
    # order: an event on the launching stream in front of every collective, one on the main stream behind each domain's
    # backward, one behind the optimiser step
    marks = []
    inner = dist.all_reduce

    def marked(t, *a, **kw):
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        marks.append(("all_reduce of %d floats" % t.numel(), e))
        return inner(t, *a, **kw)
    inner_fb = ts._fwd_bwd

    def fb(*a, **kw):
        r = inner_fb(*a, **kw)
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        marks.append(("backward of domain %d enqueued to its end" % a[2], e))
        return r
    ts._fwd_bwd = fb
    for it in range(3):
        del marks[:]
        start = torch.cuda.Event(enable_timing=True)
        start.record(torch.cuda.current_stream())
        dist.all_reduce = marked
        try:
            ts.step_all(batches)
        finally:
            dist.all_reduce = inner
        end = torch.cuda.Event(enable_timing=True)
        end.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        print("iteration %d (TrainStep.step_all, two domains, 1-rank RCCL group, FPLX_DDP_FORCE=1): %.3f ms" % (it, start.elapsed_time(end)))
        for name, e in marks:
            print("   %8.3f ms  %s" % (start.elapsed_time(e), name))
    print("buckets", len(ts.reducer.buckets), "collectives per iteration", len(ts.reducer.launched))
    dist.destroy_process_group()


def analyze(path):
    rows = []
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if r[2].startswith("adam_pack27_multi") or r[2].startswith("adam_k")]
    # iterations: between consecutive shared-segment optimiser launches (adam_pack27_multi); fall back to every second adam_k
    cuts = [i for i in adam if rows[i][2].startswith("adam_pack27_multi")] or adam[::3]
    print("iteration | RCCL kernels | first RCCL start - iteration start (ms) | last backward kernel end (ms) | RCCL kernels started before it | last RCCL end (ms)")
    for a, b in zip(cuts[:-1], cuts[1:]):
        it = rows[a + 1:b]
        if not it:
            continue
        t0 = it[0][0]
        rccl = [r for r in it if "nccl" in r[2].lower() or "rccl" in r[2].lower()]
        bwd = [r for r in it if r[2].startswith("stem_wgrad") or "wgrad" in r[2]]
        if not rccl or not bwd:
            continue
        last_bwd = max(r[1] for r in bwd)
        before = [r for r in rccl if r[0] < last_bwd]
        print("%9d | %12d | %10.3f | %10.3f | %4d of %d | %10.3f" % (a, len(rccl), (rccl[0][0] - t0) / 1e6, (last_bwd - t0) / 1e6,
                                                                   len(before), len(rccl), (max(r[1] for r in rccl) - t0) / 1e6))


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "run":
        run()
    elif len(sys.argv) >= 3 and sys.argv[1] == "analyze":
        analyze(sys.argv[2])
    else:
        print(__doc__)
