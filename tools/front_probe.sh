# A/B of the dispatch order of the row kernel's work items (kernel alone, tools/phase_probe.py): default (two fronts from
# the limb where the frame is of that kind), the older side-first order (AMT_ITEM_ORDER=2), and fixed front ratios
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
for cfg in "default" "AMT_ITEM_ORDER=2" "AMT_ITEM_ORDER=4 AMT_FRONT_RATIO=1:1" "AMT_ITEM_ORDER=4 AMT_FRONT_RATIO=2:1" "AMT_ITEM_ORDER=4 AMT_FRONT_RATIO=3:2" "AMT_ITEM_ORDER=4 AMT_FRONT_RATIO=1:2"; do
  echo "== $cfg"
  if [ "$cfg" = "default" ]; then python3 $R/tools/phase_probe.py 2>/dev/null | grep -v amdgpu.ids; else env $cfg python3 $R/tools/phase_probe.py 2>/dev/null | grep "bench frame"; fi
done
