cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03g
E2E_TIMING_ONLY=1 E2E_GENOME=1 E2E_THREADS=16/4 timeout 400 python scripts/e2e_big.py 8 4375000 /dev/shm/sta_e2e30 > gpurun_out/r03g/e2e_30x_b.log 2>&1; tail -7 gpurun_out/r03g/e2e_30x_b.log | cut -c1-420
STA_WINDOW_TRACE=1 STA_IO_THREADS=16 STA_STAGE_THREADS=4 samtools_amd/bin/samtools-amd mpileup -B -f /dev/shm/sta_e2e30/big.fa /dev/shm/sta_e2e30/big.bam 2>&1 >/dev/null | grep window | head -12
for k in 1 2 3; do STA_IO_THREADS=16 STA_STAGE_THREADS=4 STA_DRIVER_TIMING=1 samtools_amd/bin/samtools-amd mpileup -f /dev/shm/sta_e2e30/big.fa /dev/shm/sta_e2e30/big.bam 2>&1 >/dev/null | tail -1 | cut -c1-330; done
STA_DEV_THREADS=1 STA_IO_THREADS=16 STA_STAGE_THREADS=4 STA_DRIVER_TIMING=1 samtools_amd/bin/samtools-amd mpileup -f /dev/shm/sta_e2e30/big.fa /dev/shm/sta_e2e30/big.bam 2>&1 >/dev/null | tail -1 | cut -c1-330
