python tools/par_scan_compare.py 256 200 0.25 16
