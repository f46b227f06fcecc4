#!/bin/bash
N=8 timeout 300 python tools/si_graph_debug.py 2>&1 | grep -v "Warning\|warn\|detach\|print(" | tail -8 | cut -c1-420
