bash tools/gpu/pmc_sq.sh r02_ae_736 --size 736 --batch 16 --math bf16s 2>&1 | grep -E "exit|igemm_bf16|wgrad_bf16" | cut -c260-520
bash tools/gpu/pmc_sq.sh r02_ae_368 2>&1 | grep -E "igemm_kernel<64, 64|wgrad_kernel<128, 128, 0" | cut -c260-520
