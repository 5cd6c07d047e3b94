#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
export S3_BINS=1048576 PMC_GROUPS="clk mfma tcc"
EPG_S3_SYRK= bash tools/pmc_s3.sh syrk_default > gpurun_out/pmc_syrk_default.txt 2>&1
EPG_S3_SYRK=pp bash tools/pmc_s3.sh syrk_pp > gpurun_out/pmc_syrk_pp.txt 2>&1
grep -A16 "k_s3_syrk" gpurun_out/pmc_syrk_default.txt | head -40
grep -A16 "k_s3_syrk" gpurun_out/pmc_syrk_pp.txt | head -40
