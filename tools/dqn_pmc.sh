cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/$n -o r -- python tools/dqn_phase_profile.py ${2:-4096} > $O/$n.txt 2>&1; echo "$n rc $?"
  python tools/rocprof_summary.py $(find $O/$n -name "*_results.db" | head -1) 2>&1 | grep -E "mrx_k_cim_dqn|mrx_k_cim_step" | cut -c1-300 | tee -a $O/summary.txt
  rm -rf $O/$n
done
