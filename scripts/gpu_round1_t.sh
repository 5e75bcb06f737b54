for m in head fwdbwd; do
echo "=== $m"; timeout 180 python scripts/debug_graph.py $m 256 2>&1 | grep -v "amdgpu.ids" | tail -4
done
