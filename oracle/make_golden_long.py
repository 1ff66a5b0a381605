#!/usr/bin/env python3
"""oracle/make_golden_long.py -- TEST INFRASTRUCTURE. Digests of LONG outputs of the unmodified
reference CLI (oracle/_ref/hacktv_ref), written to tests/golden/ref_long.json:

  * the metric configuration (`-m i -s 16000000 --filter test`, sound on) over whole bench blocks:
    cumulative sha256 after 37 and after 128 frames -- what bench.py's parity gate and the batch-128 /
    batch-37 GPU tests compare with when the reference binary is not at hand;
  * BASELINE config 4 as written (`-m l -s 16000000 --filter --teletext demo.tti`) with the wall clock
    pinned by oracle/_ref/pin_time.so and TZ=UTC, 3 frames.

Run in the build container (needs /root/reference for oracle/_ref)."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


def ref_stream(flags, nbytes, pin=False):
    env = dict(os.environ)
    if pin:
        env["LD_PRELOAD"] = os.path.join(REF, "pin_time.so")
        env["TZ"] = "UTC"
        env.pop("HVK_PIN_TIME", None)
    p = subprocess.Popen([os.path.join(REF, "hacktv_ref")] + flags + ["-o", "-", "test"], stdout=subprocess.PIPE,
                         stderr=subprocess.DEVNULL, env=env)
    left = nbytes
    while left > 0:
        chunk = p.stdout.read(min(left, 1 << 22))
        if not chunk:
            raise RuntimeError("reference ended early")
        left -= len(chunk)
        yield chunk
    p.kill()
    p.wait()


def cumulative(flags, frame_bytes, marks, pin=False):
    h = hashlib.sha256()
    out = {}
    done = 0
    marks = sorted(marks)
    for chunk in ref_stream(flags, frame_bytes * marks[-1], pin):
        while chunk:
            nxt = next(m for m in marks if m * frame_bytes > done) * frame_bytes
            take = min(len(chunk), nxt - done)
            h.update(chunk[:take])
            done += take
            chunk = chunk[take:]
            if done == nxt:
                out[str(done // frame_bytes)] = h.copy().hexdigest()
    return out


def main():
    res = {
        "i_full": {"flags": ["-m", "i", "-s", "16000000", "--filter"], "frame_bytes": 2560000,
                   "sha256_at_frames": cumulative(["-m", "i", "-s", "16000000", "--filter"], 2560000, [1, 25, 37, 128, 165])},
        "l_tti": {"flags": ["-m", "l", "-s", "16000000", "--filter", "--teletext", "@REF@/demo.tti"], "frame_bytes": 2560000,
                  "pinned_time": 1700000000,
                  "sha256_at_frames": cumulative(["-m", "l", "-s", "16000000", "--filter", "--teletext", os.path.join(REF, "demo.tti")],
                                                 2560000, [1, 2, 3, 5], pin=True)},
    }
    with open(os.path.join(ROOT, "tests", "golden", "ref_long.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    sys.exit(main())
