mkdir -p gpurun_out
(timeout 600 python -m pytest tests -m gpu -q --tb=short --no-header -p no:cacheprovider -k "bit_exact or full_size or evaluator or empty" 2>&1 | tail -8
for w in 8 4; do for f in 0 2; do echo "waves=$w dbg=$f"; KGE_LP_WAVES=$w KGE_DBG=$f python tools/kbench.py; done; done
for w in 8; do echo "waves=$w"; KGE_LP_WAVES=$w python tools/kbench.py --what scores;  KGE_LP_WAVES=$w python tools/kbench.py --mode dot --d 400 --N 14951 --B 16384; KGE_LP_WAVES=$w python tools/kbench.py --mode complex --N 40943 --B 3134; KGE_LP_WAVES=$w python tools/kbench.py --mode dot --d 1024 --N 20000 --B 8192; KGE_LP_WAVES=$w python tools/kbench.py --B 512; done
) 2>&1 | grep -v amdgpu.ids > gpurun_out/kbench7.log
cat gpurun_out/kbench7.log
