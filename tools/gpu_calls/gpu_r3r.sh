#!/bin/bash
# SQ counters of k_ssim: round-3 kernel (default build) and its predecessor (libssrhip_ssim0.so)
mkdir -p gpurun_out/r3r
bash tools/pmc_cmd.sh ssim_new k_ssim -- env _ONE=1 python tools/exp_ssim.py > gpurun_out/r3r/pmc_new.txt 2>&1
bash tools/pmc_cmd.sh ssim_old k_ssim -- env _ONE=1 SSR_DEV_LIB=tools/_build/libssrhip_ssim0.so python tools/exp_ssim.py > gpurun_out/r3r/pmc_old.txt 2>&1
find gpurun_out/pmc_ssim_new gpurun_out/pmc_ssim_old -name "*.csv" -delete
