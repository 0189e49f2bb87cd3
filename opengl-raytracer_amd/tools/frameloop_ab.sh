# Frame-by-frame loop (one rt_raytrace_frame call per frame) under the single-frame launch knobs.
for fb in 2 1; do for ch in 1 4 16; do
printf "blocks/CU %s chunk %2s: " $fb $ch
RT_TUNE_FRAME_BLOCKS=$fb RT_TUNE_FRAME_CHUNK=$ch timeout -k 10 120 python -u opengl-raytracer_amd/tools/prof_render.py --scene ${1:-dragon80k} --frames 128 --warmup 32 --per-frame 2>&1 | grep -E "ms per frame|rror" | sed 's/.*: //'
done; done
