cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for p in 97 99 101 103; do
  if [ $p = 101 ]; then unset PB3D_LIB_PATH; else export PB3D_LIB_PATH=$R/part-based-3d-reconstruction_amd/pb3d/libpb3d_sp$p.so; fi
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/steptrace/p$p -- python3 $R/tools/slicedbench.py --shapes 1024x1024x1024 --intervals 5 --rounds 1 --reps 3 > $R/gpurun_out/steptrace/p$p.log 2>&1
done
