#!/bin/bash
# tools/ab_libs.sh [name ...] -- the metric's timed loop with the library as built ("") and with hacktv_amd/libhvk_<name>.so
# (variant builds made beside it: make -C hacktv_amd/csrc VARIANT="-D..." B=/tmp/build_x OUT=../libhvk_x.so; `base`: the
# commit before), alternating, three rounds, on ONE box: tools/ab_direct.py's lines.
names=${@:-base}
for i in 1 2 3; do
	for n in $names; do python tools/ab_direct.py "HVK_LIB=$PWD/hacktv_amd/libhvk_$n.so"; done
	python tools/ab_direct.py ""
done
