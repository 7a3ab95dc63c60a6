#!/bin/bash
# MultilinearKZG::open against the level tables, by window width (ZKHIP_LEVEL_TABLE_DELTA: widest window = log2(level) - delta); sizes as arguments
for d in 0 -1 -2 -3; do echo "delta $d"; ZKHIP_LEVEL_TABLE_DELTA=$d PERF_OPEN_MODES=tables timeout 100 python tools/perf_open.py "$@" 2>&1 | grep "^open"; done
PERF_OPEN_MODES=cached timeout 100 python tools/perf_open.py "$@" 2>&1 | grep "^open"
