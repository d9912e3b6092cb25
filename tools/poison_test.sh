R=${GRAFT_REPO_ROOT:-$(pwd)}
export DET_ONLY96=1
for cfg in '{"layers": 2, "filters": 8, "min_filters": 8, "use_nin": False, "reconstruct_filters": 8}' '{}'; do
  for opts in '"winograd": 0, "nin_gemm": 0' '"split16": 0' '"split16": 1'; do
    for p in 0 1 2 3; do
      DET_MODES=$(echo $opts | grep -q '"split16": 1' && echo 1 || echo 0) DET_CFG="$cfg" DET_OPTS="{$opts, \"debug_poison\": $p}" python $R/tools/determinism_check.py 96 3 "poison$p|$opts|$(echo $cfg | cut -c1-14)" 2>&1 | grep split16 | cut -c1-60,150-260
    done
  done
done
