cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2d; mkdir -p $O
bash scripts/gpu_sq.sh mpileup30_B r2d/sq_m30B > $O/sq_m30B.txt 2>&1; tail -3 $O/sq_m30B.txt
bash scripts/gpu_sq.sh depth30 r2d/sq_d30 > $O/sq_d30.txt 2>&1; tail -3 $O/sq_d30.txt
STA_MPLP_LEGACY=1 bash scripts/gpu_sq.sh mpileup30_B r2d/sq_m30B_legacy > $O/sq_m30B_legacy.txt 2>&1; tail -3 $O/sq_m30B_legacy.txt
