mkdir -p gpurun_out/r06
(
bash tools/ab_env.sh "" KGE_HS_NT=4 KGE_HS_NT=3
bash tools/ab_env.sh "--workload complex_wn18rr" KGE_HS_NT=4 KGE_HS_NT=3
bash tools/ab_env.sh "--workload distmult_fb15k" KGE_HS_NT=4 KGE_HS_NT=3
) 2>&1 | grep -v amdgpu > gpurun_out/r06/hs_nt4_in_situ.txt
cat gpurun_out/r06/hs_nt4_in_situ.txt
