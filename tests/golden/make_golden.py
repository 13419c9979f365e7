#!/usr/bin/env python3
"""Writes tests/golden/rng_kat.json: known answers for the reference's generator
(/root/reference/src/rng.rs:9-41, wyrand seeded with 0x0ef6f79ed30ba75a),
computed here with Python's arbitrary-precision integers straight from the
definition, independently of every C implementation in this repository.

  u64():  s += 0x2d358dccaa6c78a5 (mod 2^64); t = s * (s ^ 0x8bb84b93962eacc9)
          (128-bit); return lo64(t) ^ hi64(t)
  u32():  low 32 bits of u64()
  f32():  from_bits(0x3F800000 + (u32() >> 9)) - 1.0

The reference ships no vectors of its own; the first six match the values
quoted in this repository's survey of the reference.
"""
import json
import os
import struct

SEED = 0x0EF6F79ED30BA75A
C0 = 0x2D358DCCAA6C78A5
C1 = 0x8BB84B93962EACC9
M = (1 << 64) - 1


def u64(state):
    s = (state + C0) & M
    t = s * (s ^ C1)
    return s, (t & M) ^ (t >> 64)


def f32_bits(u):
    bits = 0x3F800000 + ((u & 0xFFFFFFFF) >> 9)
    f = struct.unpack("<f", struct.pack("<I", bits))[0] - 1.0
    return struct.unpack("<I", struct.pack("<f", f))[0]


def main():
    out = {"seed": hex(SEED), "draws": [], "jump": []}
    s = SEED
    for _ in range(64):
        s, v = u64(s)
        out["draws"].append({"u64": hex(v), "f32_bits": hex(f32_bits(v))})
    # state after n draws is seed + n*C0: spot values far into the sequence
    for n in (1, 1000, 2 ** 32 + 7, 2 ** 63 + 12345):
        s0 = (SEED + n * C0) & M
        _, v = u64(s0)
        out["jump"].append({"draws_before": str(n), "state": hex(s0), "next_u64": hex(v)})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rng_kat.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
