timeout 60 python -u scripts/debug_graph.py sync 2>&1 | tail -25
echo ----
timeout 60 python -u scripts/debug_graph.py nosync_same 2>&1 | tail -8
echo ----
timeout 60 python -u scripts/debug_graph.py nosync 2>&1 | tail -8
