#!/usr/bin/env python3
"""oracle/make_golden_hour.py -- TEST INFRASTRUCTURE. BASELINE config 5 on one device: the reference CLI's
(oracle/_ref/hacktv_ref) output of `-m i -s 16000000 --filter test` over ONE HOUR of signal -- 90 000 frames,
57.6 G samples, 230.4 GB -- reduced to what a test can carry (tests/golden/ref_hour.json):

  * for every block of 128 frames (703 whole blocks and one of 16 frames) the two 64-bit sums of
    oracle_sink.c:orc_block_sums() -- the device computes the same sums over its own output
    (hvk_block_sums()), so every one of the hour's samples is compared without crossing PCIe;
  * sha256 of the blocks around frames 0, 9 000, 45 000 and 90 000 alone;
  * the cumulative sha256 of the whole stream after 9 000, 45 000 and 90 000 frames.

Two cases: sound on (FM mono + NICAM: the metric configuration) and --noaudio. About 13 minutes of the
reference each; run `make_golden_hour.py i_hour` or `... i_hour_noaudio` (both when no case is named).
Run in the build container (needs /root/reference for oracle/_ref)."""
import ctypes
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
FRAME_BYTES = 2560000
BLOCK = 128
FRAMES = 90000
MARKS = [9000, 45000, 90000]
CASES = {
    "i_hour": ["-m", "i", "-s", "16000000", "--filter"],
    "i_hour_noaudio": ["-m", "i", "-s", "16000000", "--filter", "--noaudio"],
}


def run(flags, frames=FRAMES):
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    lib.orc_block_sums.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.POINTER(ctypes.c_uint64)]
    lib.orc_block_sums.restype = None
    nblocks = (frames + BLOCK - 1) // BLOCK
    sha_blocks = sorted({0, nblocks - 2, nblocks - 1} | {m // BLOCK for m in MARKS if m < frames})
    p = subprocess.Popen([os.path.join(REF, "hacktv_ref")] + flags + ["-o", "-", "test"], stdout=subprocess.PIPE,
                         stderr=subprocess.DEVNULL, bufsize=0)
    cum = hashlib.sha256()
    out = {"flags": flags, "frame_bytes": FRAME_BYTES, "block_frames": BLOCK, "frames": frames,
           "sums": [], "sha256_of_block": {}, "sha256_at_frames": {}}
    buf = bytearray(BLOCK * FRAME_BYTES)
    t0 = time.time()
    for b in range(nblocks):
        nf = min(BLOCK, frames - b * BLOCK)
        view = memoryview(buf)[:nf * FRAME_BYTES]
        got = 0
        while got < len(view):
            n = p.stdout.readinto(view[got:])
            if not n:
                raise RuntimeError("reference ended early")
            got += n
        # cumulative digests at the marks (a mark may fall inside a block)
        done = b * BLOCK
        pos = 0
        for m in MARKS:
            if done < m <= done + nf:
                cum.update(view[pos:(m - done) * FRAME_BYTES])
                pos = (m - done) * FRAME_BYTES
                out["sha256_at_frames"][str(m)] = cum.copy().hexdigest()
        cum.update(view[pos:])
        s = (ctypes.c_uint64 * 2)()
        lib.orc_block_sums((ctypes.c_char * len(view)).from_buffer(buf), len(view) // 4, s)
        out["sums"].append(["%016x" % s[0], "%016x" % s[1]])
        if b in sha_blocks:
            out["sha256_of_block"][str(b)] = hashlib.sha256(view).hexdigest()
        if b % 50 == 0:
            sys.stderr.write("block %d / %d, %.0f s\n" % (b, nblocks, time.time() - t0))
    p.kill()
    p.wait()
    out["reference_seconds"] = round(time.time() - t0, 1)
    return out


def main():
    names = sys.argv[1:] or list(CASES)
    path = os.path.join(ROOT, "tests", "golden", "ref_hour.json")
    for name in names:
        res = run(CASES[name])
        # (two cases may be generated side by side: merge under a lock-free re-read)
        try:
            with open(path) as f:
                allres = json.load(f)
        except FileNotFoundError:
            allres = {}
        allres[name] = res
        with open(path + ".tmp." + name, "w") as f:
            json.dump(allres, f, indent=0, sort_keys=True)
        os.replace(path + ".tmp." + name, path)
    return 0


if __name__ == "__main__":
    sys.exit(main())
