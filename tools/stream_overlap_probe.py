#!/usr/bin/env python3
"""Do two streams of captured PCT passes overlap on this GPU?  One stream replaying a forward + input-gradient graph N times
against two streams replaying one such graph each N times (2 N passes): if the second stream were free the ratio would be 1,
if the two serialise it is 2.   gpurun -- python tools/stream_overlap_probe.py"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hit_adv_amd.Dataset.synthetic import synth_batch  # noqa: E402
from hit_adv_amd.model import _sampling  # noqa: E402
from hit_adv_amd.model.pct import Pct  # noqa: E402


def make_body(model, x, feed, copies=0):
    out = {}
    spare = [torch.empty_like(x) for _ in range(copies)]

    def body():
        feed.seek(0)
        xi = x.detach().requires_grad_()
        with _sampling.using(feed):
            logits = model(xi)
        out['g'], = torch.autograd.grad(logits.logsumexp(1).sum(), xi)
        for t in spare:  # contiguous same-dtype copy_ = hipMemcpyAsync = a memcpy NODE in the captured graph
            t.copy_(out['g'])
    return body


def capture_all(bodies, streams):
    """All eager passes first, all captures last (a replay that follows eager work issued after a capture can fault on this
    stack: DESIGN.md section 5)."""
    for body, s in zip(bodies, streams):
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            body()
            body()
    torch.cuda.synchronize()
    graphs = []
    for body, s in zip(bodies, streams):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            body()
        graphs.append(g)
    return graphs


def timed(jobs, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for g, s in jobs:
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def timed_batched(jobs, n):
    """The same replays queued attack by attack (all of one stream's, then all of the next's) -- the order a host that runs
    one attack's loop after the other produces."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []
    for g, s in jobs:
        with torch.cuda.stream(s):
            for _ in range(n):
                g.replay()
        marks.append(round(time.perf_counter() - t0, 4))
    torch.cuda.synchronize()
    return time.perf_counter() - t0, marks


def main():
    res = {}
    # eager: FPS (32 workgroups, a serial chain) beside a GEMM that wants the chip
    from hit_adv_amd import ops
    data, _ = synth_batch(32, 1024)
    pts = data[:, :, :3].contiguous().cuda()
    start = torch.zeros(32, dtype=torch.long, device='cuda')
    a, b = torch.randn(32768, 256, device='cuda'), torch.randn(256, 256, device='cuda')
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def fps_job():
        with torch.cuda.stream(s1):
            for _ in range(20):
                ops.fps_pct(pts, 512, start)

    def gemm_job():
        with torch.cuda.stream(s2):
            for _ in range(400):
                torch.mm(a, b)

    def run(*fs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in fs:
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3
    run(fps_job, gemm_job)
    res['eager'] = dict(fps_ms=round(run(fps_job), 3), gemm_ms=round(run(gemm_job), 3), both_ms=round(run(fps_job, gemm_job), 3))
    for tables_ahead, copies in ((False, 0), (False, 7)):
        Pct.tables_ahead = tables_ahead
        torch.manual_seed(0)
        m = Pct(argparse.Namespace(dropout=0.2), output_channels=40).eval().cuda()
        bodies, streams = [], []
        for i in range(3):
            data, _ = synth_batch(32, 1024, first=100 * i)
            x = data[:, :, :3].transpose(1, 2).contiguous().cuda()
            streams.append(torch.cuda.Stream())
            bodies.append(make_body(m, x, _sampling.feed_for(m, 32, 1024, 1, 'cuda'), copies))
        jobs = list(zip(capture_all(bodies, streams), streams))
        n = 60
        timed(jobs[:1], 5)
        one = timed(jobs[:1], n)
        two = timed(jobs[:2], n)
        three = timed(jobs[:3], n)
        batched, marks = timed_batched(jobs[:3], n)
        res['tables_ahead=%s, memcpy nodes per pass=%d' % (tables_ahead, copies)] = dict(ms_per_pass_one_stream=round(one / n * 1e3, 3), two_streams_over_one=round(two / one, 3),
                                                     three_streams_over_one=round(three / one, 3),
                                                     three_streams_queued_one_after_the_other_over_one=round(batched / one, 3),
                                                     host_done_queueing_each_stream_at_s=marks, one_stream_total_s=round(one, 4))
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
