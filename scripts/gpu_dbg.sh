cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03g
E2E_QUICK=1 E2E_THREADS=16/4,24/4 timeout 400 python scripts/e2e_big.py 8 4375000 /dev/shm/sta_e2e30 > gpurun_out/r03g/e2e_30x.log 2>&1; tail -12 gpurun_out/r03g/e2e_30x.log | cut -c1-420
for n in 1 2; do echo "STA_DEV_THREADS=$n"; STA_DEV_THREADS=$n STA_IO_THREADS=16 STA_STAGE_THREADS=4 STA_DRIVER_TIMING=1 samtools_amd/bin/samtools-amd mpileup -f /dev/shm/sta_e2e30/big.fa /dev/shm/sta_e2e30/big.bam 2>&1 >/dev/null | tail -1 | cut -c1-400; STA_DEV_THREADS=$n STA_IO_THREADS=16 STA_STAGE_THREADS=4 STA_DRIVER_TIMING=1 samtools_amd/bin/samtools-amd mpileup -B -f /dev/shm/sta_e2e30/big.fa /dev/shm/sta_e2e30/big.bam 2>&1 >/dev/null | tail -1 | cut -c1-400; done
