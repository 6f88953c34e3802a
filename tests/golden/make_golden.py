#!/usr/bin/env python3
"""Generates the committed golden vectors.  Run from the repo root: python tests/golden/make_golden.py

lz4_kat.json     known-answer LZ4F streams produced by the system liblz4.so.1 (1.9.3) with the reference's
                 preferences (reference src/compressor.rs:138-144) for write sequences shaped like the reference's
                 packets (src/compressor.rs:126-129): 8 B, 8 B, metadata, image.
example_positions.json   instance offsets of the quads example at frames 1, 2, 60, 600: the f32 recurrence of
                 reference examples/quads_main_thread.rs:122-141.
oracle_frames.json       sha256 of the oracle's output for the quads example (regression pin of the oracle itself;
                 there is no reference-rendered image to compare with).
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
HERE = os.path.dirname(os.path.abspath(__file__))


def kat_inputs():
    """name -> list of writes (bytes)."""
    rng = np.random.default_rng(1234)
    meta = bytes([0, 1, 4, 4, 2, 16, 0xfb, 0, 1, 1, 0xfb, 0, 4, 0xfb, 0, 4])        # a bincode-like 16-byte record
    def image(n, kind):
        if kind == "flat":
            return bytes([0, 0, 0, 255]) * (n // 4)
        if kind == "noise":
            return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        out = bytearray()
        while len(out) < n:
            out += rng.integers(0, 256, 4, dtype=np.uint8).tobytes() * int(rng.integers(1, 300))
        return bytes(out[:n])
    def packet(img):
        plen = 16 + len(meta) + len(img)
        w = [plen.to_bytes(8, "big"), len(meta).to_bytes(8, "big"), meta]
        if img:
            w.append(img)
        return w
    cases = {
        "metadata_only": packet(b""),
        "tiny_image": packet(image(1024, "runs")),
        "one_block_plus_one": packet(image(65536 + 4, "runs")),
        "flat_three_blocks": packet(image(3 * 65536, "flat")),
        "noise_incompressible": packet(image(70000, "noise")),
        "two_frames": packet(image(100000, "runs")) + packet(image(100000, "runs")),
    }
    return cases


def main():
    from oracle import loader as O
    import scenes
    out = {"liblz4_version": O.liblz4().LZ4_versionNumber(), "cases": {}}
    for name, writes in kat_inputs().items():
        s = O.LibLz4Stream(0)
        for w in writes:
            s.write(w)
        stream = s.finish()
        entry = {"input_sha256": hashlib.sha256(b"".join(writes)).hexdigest(), "write_sizes": [len(w) for w in writes],
                 "stream_len": len(stream), "stream_sha256": hashlib.sha256(stream).hexdigest()}
        if len(stream) < 3000:
            entry["stream_hex"] = stream.hex()
        out["cases"][name] = entry
    with open(os.path.join(HERE, "lz4_kat.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)

    pos = {str(k): [float(v) for v in scenes.example_positions(k).reshape(-1)] for k in (1, 2, 60, 600)}
    with open(os.path.join(HERE, "example_positions.json"), "w") as f:
        json.dump(pos, f, indent=1, sort_keys=True)

    frames = {}
    for k in (1, 2, 60, 600):
        o = scenes.draw_oracle(scenes.example_scene(k, record=True), threads=1)
        frames[str(k)] = {"screen_bgra_sha256": hashlib.sha256(o["color"].tobytes()).hexdigest(),
                          "recording_rgba_sha256": hashlib.sha256(o["recording"].tobytes()).hexdigest(),
                          "fragments": o["fragments"]}
    with open(os.path.join(HERE, "oracle_frames.json"), "w") as f:
        json.dump(frames, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
