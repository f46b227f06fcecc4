#!/bin/bash
# in-graph per-dispatch traces under several conv configuration policies (one gpurun call)
export MI_CONV_TUNE=0
MI355_LIB=$PWD/yolov7_d2_amd/libA.so tools/gpu_trace.sh A > /dev/null
MI_CONV_OLDCFG=1 tools/gpu_trace.sh P0 > /dev/null
MI_CONV_MAXKC=64 MI_CONV_OCC=0 tools/gpu_trace.sh P1 > /dev/null
MI_CONV_MAXKC=128 MI_CONV_OCC=0 tools/gpu_trace.sh P2 > /dev/null
MI_CONV_MAXKC=64 MI_CONV_OCC=1 tools/gpu_trace.sh P3 > /dev/null
MI_CONV_MAXKC=64 MI_CONV_OCC=1 MI_CONV_LDSCAP=53 tools/gpu_trace.sh P4 > /dev/null
MI_CONV_MAXKC=64 MI_CONV_OCC=0 MI_CONV_LDSCAP1=80 tools/gpu_trace.sh P5 > /dev/null
MI_CONV_MAXKC=64 MI_CONV_OCC=1 MI_CONV_T=0.3 tools/gpu_trace.sh P6 > /dev/null
MI_CONV_MAXKC=128 MI_CONV_OCC=1 MI_CONV_LDSCAP=64 tools/gpu_trace.sh P7 > /dev/null
for t in A P0 P1 P2 P3 P4 P5 P6 P7; do tail -1 gpurun_out/trace_$t.log | cut -c1-200; done
