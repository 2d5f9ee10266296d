#!/bin/bash
# Re-measure the SHIPPED tile configurations (pmf_amd/tuned/gfx950.txt) on an MI355X: every BASELINE configuration is built once
# under PMF_AUTOTUNE=live (the shipped file is ignored, every conv shape is timed with every candidate, 20 repetitions each) and
# the choices accumulate in one file.  Usage (GPU box): tools/make_tune_cache.sh gpurun_out/tuned_gfx950.txt ; then copy the
# file to pmf_amd/tuned/gfx950.txt and commit it.
set -e
out=${1:-gpurun_out/tuned_gfx950.txt}
rm -f "$out"
export PMF_AUTOTUNE=live PMF_TUNE_CACHE="$out" PMF_TUNE_REPS=20
q="--steps 2 --warmup 1 --no-parity --no-cpu-baseline --no-roofline --no-f32-ref"
python bench.py $q > /dev/null                                                      # configs[2]  PMF-R34 64x2048 bs 2 (S_A)
python bench.py $q --mode infer > /dev/null                                         # configs[1]  eval bs 4 + KNN
python bench.py $q --height 480 --width 640 > /dev/null                             # S_B train
python bench.py $q --height 480 --width 640 --mode infer > /dev/null                # S_B eval bs 4
python bench.py $q --backbone resnet50 --nclasses 17 --height 32 --width 1024 > /dev/null   # configs[3] family
python bench.py $q --backbone resnet50 --nclasses 17 --height 480 --width 640 > /dev/null   # configs[3] at its RGB size
python bench.py $q --backbone resnet50 --nclasses 17 --height 512 --width 640 --mode infer > /dev/null   # S_G eval bs 4
python bench.py $q --model epmf > /dev/null                                         # configs[4]
python bench.py $q --model epmf --height 320 --width 1280 > /dev/null               # EPMF's native size
python bench.py $q --model salsanext > /dev/null                                    # f-2
python bench.py $q --height 256 --width 1024 > /dev/null                            # KITTI training crop (S_C)
wc -l "$out"
