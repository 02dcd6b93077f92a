#!/bin/bash
# usage: try_some.sh <lib> fixture...
export HEVCDL_LIB=$1; shift
for f in "$@"; do timeout 40 python tools/scratch/one_gold.py /root/repo/tests/golden/$f.npz > /tmp/o.txt 2>&1; rc=$?; echo "$f rc=$rc $(grep -E 'RESULT|fault|HSA_STATUS|exception' /tmp/o.txt | head -3)"; done
