#!/bin/bash
# A/B runs inside ONE gpurun call (boxes of the pool differ by +-4 %, one box drifts by +-3 % between runs: alternate, repeat, read
# the medians).  One script for what used to be sixteen (ab_*.sh, abl*.sh, cmp_*.sh, mel*_abl.sh, stamps_x.sh, ...):
#
#   tools/ab.sh env VAR v1 v2 ...        bench.py under VAR=v1, VAR=v2, ... (REPS alternations, default 2): value, stage times and the
#                                        per-fused-block times (us per 1 000 segments), then the median per value
#   tools/ab.sh lib [bench.py args]      two BUILDS of the library: tools/ab/libbirda_hip_old.so against birda_amd/libbirda_hip.so
#   tools/ab.sh legs "A=1 B=2" "A=0" ... the host-fed legs of bench.py (h2d_inclusive, end_to_end) under each set of assignments
#   tools/ab.sh prof VAR v1 v2 ...       rocprofv3 --kernel-trace --stats of tools/gpu_quick_bench.py per value: per-kernel averages
#   tools/ab.sh pmc TAG VAR v1 v2 ...    FETCH_SIZE / WRITE_SIZE passes per value (tools/pmc_summary.py)
#   tools/ab.sh x CMD ...                CMD with the EXPERIMENTS build swapped in (see LIBX)
#
# Environment: REPS (2), STEPS (10), BENCH_ARGS (extra bench.py arguments), MODEL / NSEG for prof (birdnet_v24 / 1000), LIBX=1 =
# run with the EXPERIMENTS build of the library swapped in (tools/ab/libbirda_hip_x.so, built with
#   make -C birda_amd/csrc EXPERIMENTS=1 BUILD=_build_x LIB=../../tools/ab/libbirda_hip_x.so
# -- it carries the measured alternatives of mbconv_cfgs.inc, the phase clock and the A/B switches that the product build compiles
# out: BIRDA_HIP_MB_DBG / _MEL_DBG ablation bits, _LANES, _NLANES, _SUBSLICES, _FIRST_SUBSLICE, _MB_TWIN, _MB_RING, _MB_PERSIST,
# _MEL_PAIR, _EVENT_FENCE, _RESAMPLE_F32, _STEM_F32, BIRDA_HOST_PACK_SUBSLICES; kernels.hpp BH_XENV).
cd ${GRAFT_REPO_ROOT:-.}
root=$(pwd)
mode=$1; shift
swap_in() { cp birda_amd/libbirda_hip.so /tmp/libbirda_hip_keep.so; cp "$1" birda_amd/libbirda_hip.so; }
swap_out() { cp /tmp/libbirda_hip_keep.so birda_amd/libbirda_hip.so; }
[ -n "$LIBX" ] && [ "$mode" != lib ] && swap_in tools/ab/libbirda_hip_x.so
row() {   # label -> one line: value, median of five, mel, mbconv, per-block times
  python bench.py --no-cpu-baseline --no-extra-legs --steps ${STEPS:-10} --warmup 2 $BENCH_ARGS "${@:2}" 2>/tmp/ab_err.txt | python -c "
import json,sys
t=sys.stdin.read()
if not t.strip(): print('$1 | bench.py FAILED:', open('/tmp/ab_err.txt').read()[-300:].replace(chr(10),' '), file=sys.stderr); sys.exit(0)
d=json.loads(t); f=d['fused_block_us_per_1000_segments']; s=d['stage_us_per_segment']
print('$1 | %7.0f seg/s  med5 %7.0f  mel %.3f  mbconv %.3f | %s' % (d['value'], d.get('repeats',{}).get('median_of_5',0), s['mel'], s['mbconv'], ' '.join('%6.0f' % x for x in f.values())))" | tee -a /tmp/ab_rows.txt
}
medians() { python - <<'PY'
import statistics as st, collections
rows=collections.OrderedDict()
for l in open('/tmp/ab_rows.txt'):
    lab,a,b=l.split('|'); t=a.split(); rows.setdefault(lab.strip(),[]).append([float(t[0]),float(t[3]),float(t[5]),float(t[7])]+[float(x) for x in b.split()])
for lab,r in rows.items():
    m=[st.median(c) for c in zip(*r)]
    print('MEDIAN %-28s | %7.0f seg/s  med5 %7.0f  mel %.3f  mbconv %.3f | %s' % (lab, m[0], m[1], m[2], m[3], ' '.join('%6.0f' % x for x in m[4:])))
# ... and, per run, every block's time over the SAME run's mel time (a kernel no variant of the fused block touches): a box that
# changes mode between runs (+-7 %, every kernel alike: profiles/r5_h_repeat_bench.txt) moves numerator and denominator together
for lab,r in rows.items():
    q=[st.median([row[4+k]/row[2] for row in r]) for k in range(len(r[0])-4)]
    print('PER-MEL %-28s | mbconv/mel %.3f | %s' % (lab, st.median([row[3]/row[2] for row in r]), ' '.join('%6.0f' % x for x in q)))
PY
}
case $mode in
env)  var=$1; shift; rm -f /tmp/ab_rows.txt
      for rep in $(seq 1 ${REPS:-2}); do for v in "$@"; do export $var="$v"; row "$var=$v"; done; done; unset $var; medians ;;
lib)  cp birda_amd/libbirda_hip.so /tmp/libbirda_hip_new.so; rm -f /tmp/ab_rows.txt
      for rep in $(seq 1 ${REPS:-2}); do
        cp tools/ab/libbirda_hip_old.so birda_amd/libbirda_hip.so; row old "$@"
        cp /tmp/libbirda_hip_new.so birda_amd/libbirda_hip.so; row new "$@"
      done; cp /tmp/libbirda_hip_new.so birda_amd/libbirda_hip.so; medians ;;
libs) # tools/ab.sh libs tagA tagB ...: tools/ab/libbirda_hip_<tag>.so in turn (tools/build_variants.sh), REPS alternations, medians
      cp birda_amd/libbirda_hip.so /tmp/libbirda_hip_new.so; rm -f /tmp/ab_rows.txt
      for rep in $(seq 1 ${REPS:-2}); do for tag in "$@"; do cp tools/ab/libbirda_hip_$tag.so birda_amd/libbirda_hip.so; row $tag; done; done
      cp /tmp/libbirda_hip_new.so birda_amd/libbirda_hip.so; medians ;;
legs) for set in "$@"; do
        env $set python bench.py --no-cpu-baseline --steps ${STEPS:-5} --warmup 2 $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); h=d.get('h2d_inclusive',{}); e=d.get('end_to_end',{})
g=lambda x,k: round(x[k]['value']) if isinstance(x.get(k),dict) and 'value' in x[k] else None
print('[$set]', round(d['value']), 'b256', round(d.get('value_at_batch_256',0)), 'b512', round(d.get('value_at_batch_512',0)), {k:g(h,k) for k in h if g(h,k)}, {k:g(e,k) for k in e if g(e,k)})"
      done ;;
prof) var=$1; shift; cd /tmp && export TMPDIR=/tmp; i=0
      for v in "$@"; do i=$((i+1)); export $var="$v"
        rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/abprof_$i -- python3 $root/tools/gpu_quick_bench.py ${MODEL:-birdnet_v24} ${NSEG:-1000} ${NSEG:-1000} > /dev/null 2>&1
        echo "== $var=$v"; python3 $root/tools/kstats.py $root/gpurun_out/abprof_$i | sort | awk '{printf "%s %s | ", $1, $5}'; echo
      done; cd $root ;;
pmc)  tag=$1; var=$2; shift 2; cd /tmp && export TMPDIR=/tmp
      for v in "$@"; do out=$root/gpurun_out/${tag}_${var}_$v; mkdir -p $out; export $var="$v"
        rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2> $out/pmc_fetch.log
        rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2> $out/pmc_write.log
        python3 $root/tools/pmc_summary.py $out $out/traffic.json > $out/traffic.txt; echo "== $var=$v"; head -4 $out/traffic.txt; rm -rf $out/pmc_fetch $out/pmc_write
      done; cd $root ;;
x)    [ -z "$LIBX" ] && swap_in tools/ab/libbirda_hip_x.so; "$@"; [ -z "$LIBX" ] && swap_out ;;
*)    sed -n 2,22p $0 ;;
esac
[ -n "$LIBX" ] && [ "$mode" != lib ] && swap_out
exit 0
