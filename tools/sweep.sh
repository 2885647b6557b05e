#!/bin/bash
# GPU box helper: source tools/sweep.sh; EXTRA="--patch-cells 8" run tag   -> prints "tag ms/step tendency_ms patch_cells"
run() { tag=$1; shift; python bench.py --no-cpu --steps 8 --warmup 2 --tend-iters 6 $EXTRA "$@" > gpurun_out/sw.json 2> gpurun_out/sw.err && python -c "
import json;d=json.load(open('gpurun_out/sw.json'));print('$tag', round(d['ms_per_step'],3), round(d['tendency_kernel']['avg_launch_ms'],3), d['config']['patch_cells'])"; }
