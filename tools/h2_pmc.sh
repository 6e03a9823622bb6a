# usage (GPU box, repo root): bash tools/h2_pmc.sh <tag> B H W Cin Cout K [reps] [b3]  -- SQ / LDS counters of the convolution kernel, one pass per group
tag=$1; shift
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  out=gpurun_out/hpmc_${tag}_$i
  rm -rf $out && mkdir -p $out
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out -- python tools/h2_conv_one.py "$@" > $out/log.txt 2>&1 || { tail -5 $out/log.txt; }
  python - $out <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'gemm_bf16x6' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        print(f'{k}: mean {sum(v) / len(v):.4g} over {len(v)} dispatches')
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    ts = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f)) if 'gemm_bf16x6' in r['Kernel_Name']]
    if ts:
        print(f'kernel time us: mean {sum(ts) / len(ts):.1f} min {min(ts):.1f} n {len(ts)}')
PY
  rm -rf $out
  i=$((i+1))
done
