"""Experiment: does running two half-batches on two HIP streams (two model instances, own workspaces) beat one
full batch on one stream?  The tail of the forward (stage 3, head) leaves most CUs idle."""
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import btsbot_amd
from bench import CONFIG, seeded_weights
from btsbot_amd.synthetic import synthetic_batch

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2


def make():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = btsbot_amd.mm_ConvNeXt(CONFIG, precision="bf16")
    seeded_weights(m)
    return m.to(dev).eval()


img, meta, _ = synthetic_batch(B, seed=3)
img, meta = img.to(dev), meta.to(dev)
models = [make() for _ in range(NS)]
streams = [torch.cuda.Stream() for _ in range(NS)]
parts = [(img[i * B // NS:(i + 1) * B // NS].contiguous(), meta[i * B // NS:(i + 1) * B // NS].contiguous()) for i in range(NS)]


def run_split(steps):
    for _ in range(steps):
        for m, s, (im, me) in zip(models, streams, parts):
            with torch.cuda.stream(s), torch.no_grad():
                m(image_input=im, metadata_input=me)


def run_one(steps):
    with torch.no_grad():
        for _ in range(steps):
            models[0](image_input=img, metadata_input=meta)


for fn, tag in ((run_one, "one stream, full batch"), (run_split, f"{NS} streams, batch / {NS} each")):
    fn(10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(100)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{tag}: {1e3 * el / 100:.4f} ms per {B} alerts, {B * 100 / el:,.0f} alerts/s")
# also: full batches alternating between the streams (consecutive calls overlap)
def run_alt(steps):
    for i in range(steps):
        k = i % NS
        with torch.cuda.stream(streams[k]), torch.no_grad():
            models[k](image_input=img, metadata_input=meta)
run_alt(10); torch.cuda.synchronize(); t0 = time.perf_counter(); run_alt(100); torch.cuda.synchronize(); el = time.perf_counter() - t0
print(f"{NS} streams, full batches alternating: {1e3 * el / 100:.4f} ms per {B} alerts, {B * 100 / el:,.0f} alerts/s")

# ScoreStream.map as bench.py uses it, with the caller on the default stream and on a stream of its own
scorer = btsbot_amd.ScoreStream(models[0], depth=NS, inputs_ready=True)
def run_map(steps):
    o = None
    for o in scorer.map((img, meta) for _ in range(steps)):
        pass
    return o
for tag, ctx in (("default stream", None), ("own stream", torch.cuda.Stream())):
    def go(n):
        if ctx is None:
            return run_map(n)
        with torch.cuda.stream(ctx):
            return run_map(n)
    go(10); torch.cuda.synchronize(); t0 = time.perf_counter(); go(100); torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"ScoreStream.map, caller on {tag}: {1e3 * el / 100:.4f} ms per {B} alerts, {B * 100 / el:,.0f} alerts/s")

def timeit(fn, tag):
    fn(10); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(100); torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"{tag}: {1e3 * el / 100:.4f} ms per {B} alerts")
def alt_keep(steps):
    keep = []
    for i in range(steps):
        k = i % NS
        with torch.cuda.stream(streams[k]), torch.no_grad():
            keep.append(models[k](image_input=img, metadata_input=meta))
    return keep
timeit(alt_keep, "alternating, outputs kept")
def alt_events(steps):
    for i in range(steps):
        k = i % NS
        with torch.cuda.stream(streams[k]), torch.no_grad():
            models[k](image_input=img, metadata_input=meta)
            e = torch.cuda.Event(); e.record(streams[k])
timeit(alt_events, "alternating, one event record per batch, no waits")
def alt_evsync(steps):
    evs = collections.deque()
    for i in range(steps):
        k = i % NS
        with torch.cuda.stream(streams[k]), torch.no_grad():
            models[k](image_input=img, metadata_input=meta)
            e = torch.cuda.Event(); e.record(streams[k]); evs.append(e)
        if len(evs) > 4:
            evs.popleft().synchronize()
import collections
timeit(alt_evsync, "alternating, event per batch, host waits 4 batches behind")
def alt_positional(steps):
    for i in range(steps):
        k = i % NS
        with torch.cuda.stream(streams[k]), torch.no_grad():
            scorer.models[k](img, meta)
timeit(alt_positional, "alternating, the scorer's replicas")
timeit(lambda n: run_map(n), "ScoreStream.map again (end of script)")
own = torch.cuda.Stream()
def map_own(n):
    with torch.cuda.stream(own):
        return run_map(n)
timeit(map_own, "ScoreStream.map again, own stream")
scorer.streams = streams
timeit(lambda n: run_map(n), "ScoreStream.map on the script's first two streams")
scorer.streams = [torch.cuda.Stream() for _ in range(NS)]
timeit(lambda n: run_map(n), "ScoreStream.map on two fresh streams")
hi = [torch.cuda.Stream(priority=-1) for _ in range(NS)]
scorer.streams = hi
timeit(lambda n: run_map(n), "ScoreStream.map on two high-priority streams")
