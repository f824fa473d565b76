#!/bin/bash
# usage: ab.sh tag [ENV=VAL ...]
tag=$1; shift
mkdir -p gpurun_out/r2
env "$@" python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2> gpurun_out/r2/ab_$tag.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'])"
