cd $GRAFT_REPO_ROOT
for env in "X=1" "G4S_PB_PCHUNK=65536 G4S_PB_CCHUNK=32768" "G4S_PB_PCHUNK=131072 G4S_PB_CCHUNK=32768" "G4S_PB_PCHUNK=65536 G4S_PB_CCHUNK=65536" "G4S_PB_PCHUNK=49152 G4S_PB_CCHUNK=32768"; do
  echo "== $env"; env $env python3 tools/dist_probe.py --workload rmat 2>/dev/null | grep -E "rank [0-9]|slowest" | sed 's/referenced remote columns.*per step//' | cut -c1-150
done
