export TMPDIR=/tmp
python tools/variant_bench.py --variants 8,10,12,13,14,15,16,17,18,19,20,22,24 > gpurun_out/r02_k_variant_gcups.txt 2>&1 && tail -14 gpurun_out/r02_k_variant_gcups.txt
for c0 in 0.3 1.0 2.0; do for c1 in 0.3 0.535 0.8 1.1; do
  echo "c0=$c0 c1=$c1 $(PC_CHOOSE_C0=$c0 PC_CHOOSE_C1=$c1 python tools/quick_bench.py -n 3000 --steps 3 2>&1 | grep 'step 2' | sed 's/.*align \([0-9.]*\).*GCUPS \([0-9.]*\)/align \1 GCUPS \2/')" >> gpurun_out/r02_k_choose_sweep.txt
done; done
cat gpurun_out/r02_k_choose_sweep.txt
