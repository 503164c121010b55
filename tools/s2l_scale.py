#!/usr/bin/env python3
"""Head-room of config 4's strong scaling, measured on one GPU: the 65 536-window batch on this
GPU against ONE rank's 8192-window shard as run_sharded_pipelined issues it -- one launch, four pieces on one stream,
four pieces on two alternating streams (two engines, each with its own workspace) -- and the screen alone
(thresh = 10: no period is ever flagged).  The full batch is the shard's windows eight times over (same distribution,
no 2 GiB of synthesis).  Kernel times from HIP events of the library, stream times from torch events."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pyperiod_amd import PeriodEngine  # noqa: E402
from pyperiod_amd.synth import multi_sinusoid_batch  # noqa: E402

W, REPS = 8192, 5
x = torch.from_numpy(multi_sinusoid_batch(0, W, 4096)).to("cuda:0")
xfull = x.repeat(8, 1)
engs = [PeriodEngine(0), PeriodEngine(0)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def s2l(eng, xs, thresh=0.05):
    return eng.small_to_large(xs, thresh, None, False, False, cap=32, want_bases=False, nosync=True)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(REPS):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / REPS


def pieces_one_stream(n):
    step = W // n
    for k in range(n):
        s2l(engs[0], x[k * step:(k + 1) * step])


def pieces_two_streams(n):
    """piece k on stream k % 2: the head of piece k+1 fills the tail of piece k"""
    step = W // n
    cur = torch.cuda.current_stream()
    done = []
    for k in range(n):
        s = streams[k % 2]
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            s2l(engs[k % 2], x[k * step:(k + 1) * step])
        done.append(s)
    for s in streams:
        cur.wait_stream(s)


full = timed(lambda: s2l(engs[0], xfull))
shard = timed(lambda: s2l(engs[0], x))
print(f"full batch 65536 windows: {full:8.3f} ms   ideal shard (1/8): {full / 8:6.3f} ms   target shard (full / 7.6): {full / 7.6:6.3f} ms")
print(f"shard 8192 windows, one launch: {shard:6.3f} ms   -> compute-only speed-up at 8 GPUs {full / shard:5.2f}x")
from pyperiod_amd.dist import piece_rows  # noqa: E402


def pieces_weighted(weights):
    off = 0
    for rows in piece_rows(W, weights):
        s2l(engs[0], x[off:off + rows])
        off += rows


a = timed(lambda: pieces_weighted((1, 3)))
print(f"shard as pieces (1, 3): 2048 + 6144 windows: {a:6.3f} ms ({full / a:5.2f}x)")
for n in (2, 4, 8):  # 2 = what run_sharded_pipelined issues by default
    a = timed(lambda: pieces_one_stream(n))
    b = timed(lambda: pieces_two_streams(n))
    print(f"shard as {n} pieces: one stream {a:6.3f} ms ({full / a:5.2f}x)   two alternating streams {b:6.3f} ms ({full / b:5.2f}x)")
scr = timed(lambda: s2l(engs[0], x, 10.0))
print(f"screen alone (thresh 10, 2047 passes per pair, no event): shard {scr:6.3f} ms")
scrf = timed(lambda: s2l(engs[0], xfull, 10.0))
print(f"screen alone, full batch: {scrf:8.3f} ms ({scrf / 8:6.3f} per shard)")
