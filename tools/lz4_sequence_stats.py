#!/usr/bin/env python3
"""What an LZ4F stream file is made of: blocks, sequences per block, the lengths of literal runs and matches and the
match distances, as histograms.  The compression kernel's time is per sequence, not per byte, so this says what a frame
costs.  usage: python3 tools/lz4_sequence_stats.py FILE.sz [FILE.sz ...]"""
import collections
import json
import struct
import sys


def stats(path):
    data = open(path, "rb").read()
    # (a file whose first bytes were cut off -- the capture bench drops its warm-up -- starts at a block header)
    pos = 7 if data[:4] == b"\x04\x22\x4d\x18" else 0
    blocks = stored = 0
    raw = 0
    lit_hist, match_hist, dist_hist = collections.Counter(), collections.Counter(), collections.Counter()
    n_seq = 0

    def lit_class(n):
        return "0" if n == 0 else "1-3" if n < 4 else "4-14" if n < 15 else "15-63" if n < 64 else "64-255" if n < 256 else "256+"

    def match_class(n):
        return "4" if n == 4 else "5-7" if n < 8 else "8-11" if n < 12 else "12-18" if n < 19 else "19-63" if n < 64 else "64-255" if n < 256 else "256+"

    def dist_class(d):
        return "<=4" if d <= 4 else "5-64" if d <= 64 else "65-4092" if d <= 4092 else "4093-15000" if d <= 15000 else "15001-16000" if d <= 16000 else "16001+"

    while pos + 4 <= len(data):
        (h,) = struct.unpack_from("<I", data, pos)
        pos += 4
        if h == 0:
            break
        n = h & 0x7fffffff
        blocks += 1
        if h >> 31:
            stored += 1
            raw += n
            pos += n
            continue
        end = pos + n
        out = 0
        while pos < end:
            tok = data[pos]; pos += 1
            lit = tok >> 4
            if lit == 15:
                while True:
                    b = data[pos]; pos += 1
                    lit += b
                    if b != 255:
                        break
            pos += lit
            out += lit
            lit_hist[lit_class(lit)] += 1
            if pos >= end:
                break
            dist = data[pos] | (data[pos + 1] << 8); pos += 2
            ml = tok & 15
            if ml == 15:
                while True:
                    b = data[pos]; pos += 1
                    ml += b
                    if b != 255:
                        break
            ml += 4
            out += ml
            n_seq += 1
            match_hist[match_class(ml)] += 1
            dist_hist[dist_class(dist)] += 1
        raw += out
    return {"file": path, "bytes": len(data), "raw_bytes": raw, "blocks": blocks, "stored_blocks": stored, "sequences": n_seq,
            "raw_bytes_per_sequence": raw / max(1, n_seq), "literal_runs": dict(lit_hist), "match_lengths": dict(match_hist),
            "match_distances": dict(dist_hist)}


if __name__ == "__main__":
    for p in sys.argv[1:]:
        print(json.dumps(stats(p)), flush=True)
