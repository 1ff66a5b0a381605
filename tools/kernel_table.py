#!/usr/bin/env python3
"""tools/kernel_table.py -- "configuration -> kernels launched", from the engine itself (hvk_kernel_plan(): which path a
configuration takes is decided when the engine is opened on a device). Markdown to stdout; DESIGN.md section 2 carries
the output of the round's last run. Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hacktv_amd as H

F = H.FLAG_FILTER
ROWS = [
    ("BASELINE config 1: `-m pal -s 16000000`", "pal", 16000000, 0, 0, {}),
    ("BASELINE config 2 (the metric): `-m i -s 16000000 --filter`", "i", 16000000, F, 0, {}),
    ("... `--noaudio`", "i", 16000000, F | H.FLAG_NOAUDIO, 0, {}),
    ("BASELINE config 3: `-m m -s 13500000 --filter`", "m", 13500000, F, 0, {}),
    ("BASELINE config 4: `-m l -s 16000000 --filter --teletext ...`", "l", 16000000, F, 0, {"teletext": 1}),
    ("`-m l --filter` (SECAM, no inserters)", "l", 16000000, F, 0, {}),
    ("`-m secam --secam-field-id`", "secam", 16000000, 0, 0, {"secam_field_id": 1}),
    ("`-m i --filter --vits --vitc --wss 4:3 --acp --cc608`", "i", 16000000, F, 0, {"vits": 1, "vitc": 1, "wss": 8, "acp": 1, "cc608": 1}),
    ("`-m i --filter --sis dcsis`", "i", 16000000, F, 0, {"sis": 1}),
    ("`-m pal --s-video`", "pal", 16000000, 0, 0, {"s_video": 1}),
    ("`-m i --filter --pixelrate 13500000`", "i", 16000000, F, 13500000, {}),
    ("`-m ntsc -s 16000000 --s-video --filter --pixelrate 13500000` (lines of two widths)", "ntsc", 16000000, F, 13500000, {"s_video": 1}),
    ("`-m i --filter --interlace`", "i", 16000000, F, 0, {"interlace": 1}),
    ("`-m i --filter --offset 2000000 --swap-iq`", "i", 16000000, F, 0, {"offset": 2000000, "swap_iq": 1}),
    ("`-m pal-fm -s 14000000 --filter` (FM video)", "pal-fm", 14000000, F, 0, {}),
    ("`-m i --filter --raw-bb-file ...`", "i", 16000000, F, 0, {"raw_bb": 1, "raw_bb_blanking_level": 2000, "raw_bb_white_level": 21000}),
    ("`-m 405 -s 8100000` (405 lines)", "405", 8100000, 0, 0, {}),
    ("`-m apollo-fsc -s 8000000` (field-sequential colour)", "apollo-fsc", 8000000, 0, 0, {}),
]
print("| configuration | what runs (`hvk_kernel_plan()`) |")
print("|---|---|")
for label, mode, sr, flags, pr, members in ROWS:
    conf = H.preset(mode, flags)
    for k, v in members.items():
        setattr(conf, k, v)
    try:
        with H.Engine(conf, sr, device=0, max_frames=2, pixel_rate=pr) as e:
            plan = e.kernel_plan().strip().split("\n")
    except H.HvkError as ex:
        plan = ["refused (%d)" % ex.code]
    print("| %s | %s |" % (label, "<br>".join("`%s`" % p.replace("|", "/") if False else p.replace("|", "/") for p in plan)))
