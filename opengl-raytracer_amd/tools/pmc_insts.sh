# Instruction counts of the path kernel for the product library and every library under csrc/diag/ (one PMC pass each).
# usage: bash pmc_insts.sh [scene]
S=${1:-dragon80k}
cd /tmp && export TMPDIR=/tmp
for L in /root/repo/opengl-raytracer_amd/csrc/librt_hip.so /root/repo/opengl-raytracer_amd/csrc/diag/*.so; do
  [ -f "$L" ] || continue
  export RT_HIP_LIB=$L
  O=/root/repo/gpurun_out/pmc_insts/$(basename $L .so)
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES -d $O -o p --output-format csv -- python3 /root/repo/opengl-raytracer_amd/tools/prof_render.py --scene $S --frames 8 --warmup 8 > /dev/null 2>&1
  echo "== $(basename $L)"
  (cd /root/repo && python3 opengl-raytracer_amd/tools/rocprof_summary.py $O | grep k_paths | sed 's/void k_paths<false, false>(DScene, DCamera, PathsArgs)//')
done
