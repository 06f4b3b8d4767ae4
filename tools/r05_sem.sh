cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05sem
mkdir -p $O
( echo "config 2: A = overlapped (default), B = --no-overlap-semantic (encoder after the backward, serial)"; bash tools/ab_opts.sh "--config 2" "--config 2 --no-overlap-semantic" 2
  echo "config 3 (no semantic loss) for reference"; bash tools/ab_opts.sh "--config 3" "--config 3" 1 ) 2>&1 | tee $O/ab_sem.txt
# kernel summary of the serial variant: the encoder's kernels at their stand-alone durations
rm -rf $O/prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python3 bench.py --config 2 --no-overlap-semantic --no-cpu-baseline --no-also --no-kernel-events --steps 5 --warmup 2 > $O/prof.log 2>&1
cp $O/prof/k_kernel_stats.csv $O/kernel_stats_c2_serial.csv; rm -rf $O/prof
python - <<'PY'
import csv,re
rows=list(csv.DictReader(open('gpurun_out/r05sem/kernel_stats_c2_serial.csv')))
steps=7
sw=0.0
for r in rows:
    n=r['Name']
    if re.search('swin|layernorm|gemm_nt_wide_kernelIDF16bLi64ELi5|gemm_nt_wide_kernelIDF16bLi128|gemm_nt_kernelIDF16bLi0ELi1|gemm_nt_kernelIDF16bLi0ELi5|bicubic|semantic|crop|patch_embed|gelu|softmax', n):
        sw+=int(r['TotalDurationNs'])
        print(n[:70], int(r['Calls'])/steps, round(float(r['AverageNs'])/1e3,1), round(int(r['TotalDurationNs'])/steps/1e3,1))
print("encoder-ish kernels per step (us):", sw/steps/1e3)
PY
