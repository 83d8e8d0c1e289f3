bash tools/gpu/pmc_sq.sh r02_736 --size 736 --batch 16 --math bf16s
bash tools/gpu/pmc_sq.sh r02_368
