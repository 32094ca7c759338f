#!/bin/sh
# TEST INFRASTRUCTURE ONLY.  usage: trace_insert.sh POINTS SOURCE > PATCHED
# Writes SOURCE with the statements of POINTS (line|token|statement) inserted after the named lines; fails if a token is not on its line.
set -eu
points=$1
src=$2
script=$(mktemp)
trap 'rm -f "$script"' EXIT
grep -v '^#' "$points" | while IFS='|' read -r line token stmt; do
    [ -n "$line" ] || continue
    if ! sed -n "${line}p" "$src" | grep -qF -- "$token"; then
        echo "trace_insert: line $line of $src does not contain '$token' (the reference changed?)" >&2
        exit 1
    fi
    printf '%sa\\\n%s\n' "$line" "$stmt" >> "$script"
done
sed -f "$script" "$src"
