#!/bin/bash
# Sweep of the host pipe's DMA chunk size and ring depth (build/host_api_rate, see host_api_rate.cpp).
set -e
for sl in 4 8; do for c in 8192 16384 32768 65536; do
  echo "slots=$sl chunk_kb=$c"
  ZBRA_HIP_HOST_SLOTS=$sl ZBRA_HIP_HOST_CHUNK_KB=$c timeout -k 10 100 build/host_api_rate 28 26
done; done
