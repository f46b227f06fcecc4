#!/bin/bash
timeout 300 bash tools/ctx_probe.sh 2>&1 | grep -v "^W2026\|amdgpu.ids" | tail -80
