python tools/par_scan_stats.py 256 200 2>&1 | tail -5
