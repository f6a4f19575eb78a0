# needs the experiment build of the library: make -C dense-visual-odometry_amd/csrc experiments
export DVO_HIP_LIB=${DVO_HIP_LIB:-$PWD/ab_libs/libdvo_hip_exp.so}
for tb in 512 1024 2048 4096 8192; do for mg in 16 32; do echo -n "target $tb maxg $mg: "; DVO_TARGET_BLOCKS=$tb DVO_MAX_GROUPS=$mg DVO_EXP_LAUNCHES=40 python tools/exp_ablate.py 2>/dev/null; done; done
