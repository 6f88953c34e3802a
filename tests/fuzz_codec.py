#!/usr/bin/env python3
"""Randomised sweep of the GPU LZ4F compressor against the system liblz4 (what the reference links through lzzzz):
streams of 1-4 packets with random sizes around the 64 KiB block boundary and random content (noise, runs, repeats at
distances around the 64 KiB window, copies of the previous packet, mixtures), byte-identical stream by stream.

usage (on a GPU box, from the repo root): python3 tests/fuzz_codec.py [--rounds 200] [--seed 0] [--max-seconds 600]
(tests/test_gpu_fuzz.py runs a few fixed seeds under pytest)
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def random_image(rng, previous):
    n = int(rng.choice([4, 64, 4352, 65532, 65536, 65540, 131072, 3 * 65536 + 8, int(rng.integers(1, 120_000)) * 4]))
    kind = rng.choice(["noise", "flat", "runs", "far", "previous", "mix", "text"])
    if kind == "noise":
        out = rng.integers(0, 256, n, dtype=np.uint8)
    elif kind == "flat":
        out = np.tile(rng.integers(0, 256, 4, dtype=np.uint8), n // 4)
    elif kind == "runs":
        parts = []
        total = 0
        while total < n:
            k = int(rng.integers(1, 40_000)) * 4
            parts.append(np.tile(rng.integers(0, 256, 4, dtype=np.uint8), k // 4)); total += k
        out = np.concatenate(parts)[:n]
    elif kind == "far":
        # repeats at distances just below, at and above the 65535-byte offset limit
        out = rng.integers(0, 256, n, dtype=np.uint8)
        for _ in range(20):
            d = int(rng.choice([65531, 65532, 65535, 65536, 65540, 70000, 1000, 12]))
            if n > d + 64:
                dst = int(rng.integers(d, n - 32)); ln = int(rng.integers(4, min(2000, n - dst)))
                out[dst:dst + ln] = out[dst - d:dst - d + ln]
    elif kind == "previous" and previous is not None and len(previous) >= 4:
        reps = -(-n // len(previous))
        out = np.tile(np.frombuffer(previous, dtype=np.uint8), reps)[:n].copy()
        flips = rng.integers(0, n, max(1, n // 5000))
        out[flips] ^= 0x55
    elif kind == "text":
        base = rng.integers(0, 256, 3000, dtype=np.uint8)
        parts = []
        total = 0
        while total < n:
            s = int(rng.integers(0, 2900)); ln = int(rng.integers(4, 100))
            parts.append(base[s:s + ln]); total += len(parts[-1])
        out = np.concatenate(parts)[:n]
    else:
        a = rng.integers(0, 256, n, dtype=np.uint8)
        b = np.tile(rng.integers(0, 256, 8, dtype=np.uint8), n // 8 + 1)[:n]
        sel = np.repeat(rng.random(n // 512 + 1) < 0.5, 512)[:n]
        out = np.where(sel, a, b).astype(np.uint8)
    out = np.ascontiguousarray(out[: n - n % 4])
    return out.tobytes()


def run_round(seed, n_streams=16):
    from oracle import loader as O
    import test_gpu_codec as T
    rng = np.random.default_rng(seed)
    packets_per_stream = []
    for s in range(n_streams):
        packets, previous = [], None
        for _ in range(int(rng.integers(1, 5))):
            meta = rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8).tobytes()
            if rng.random() < 0.1:
                packets.append((meta, None))                  # dropped frame: metadata only
            else:
                previous = random_image(rng, previous)
                packets.append((meta, previous))
        packets_per_stream.append(packets)
    got = T.gpu_streams(packets_per_stream, n_streams)
    for s, packets in enumerate(packets_per_stream):
        want = T.reference_stream(lambda: O.LibLz4Stream(0), packets)
        if got[s] != want:
            first = next((i for i, (x, y) in enumerate(zip(got[s], want)) if x != y), min(len(got[s]), len(want)))
            sizes = [(len(m), None if i is None else len(i)) for m, i in packets]
            return f"stream {s}: {len(got[s])} bytes on the GPU, {len(want)} from liblz4, first difference at byte {first}; packets {sizes}"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-seconds", type=float, default=600.0)
    args = ap.parse_args()
    import torch  # noqa: F401  (before the renderer's library: one HIP runtime per process)
    t0 = time.time()
    done = 0
    for i in range(args.rounds):
        err = run_round(args.seed + i)
        done += 1
        if err:
            print(f"MISMATCH seed {args.seed + i}: {err}", flush=True)
            return 1
        if i % 5 == 0:
            print(f"[{i}] seed {args.seed + i} ok ({time.time() - t0:.0f} s)", flush=True)
        if time.time() - t0 > args.max_seconds:
            break
    print(f"{done} rounds of 16 random streams byte-identical to liblz4 (seeds {args.seed}..{args.seed + done - 1}, {time.time() - t0:.0f} s)", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
