#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}/tools/ubench
T="timeout 120"
for b in "$@"; do
  echo "== $b =="
  case $b in
    *st) $T ./$b 128 128 | tail -34 ;;
    *)   $T ./$b 128 128 | tail -3 ;;
  esac
done
