#!/bin/sh
# TEST INFRASTRUCTURE ONLY.  usage: trace_insert.sh POINTS SOURCE > PATCHED
# Writes SOURCE with the statements POINTS names for it (file|line|token|statement) inserted after the named lines; fails if a token is not on
# its line (a reference whose lines moved is never patched in the wrong place).
set -eu
points=$1
src=$2
name=$(basename "$src")
script=$(mktemp)
trap 'rm -f "$script"' EXIT
grep -v '^#' "$points" | while IFS='|' read -r file line token stmt; do
    [ "$file" = "$name" ] || continue
    if ! sed -n "${line}p" "$src" | grep -qF -- "$token"; then
        echo "trace_insert: line $line of $src does not contain '$token' (the reference changed?)" >&2
        exit 1
    fi
    printf '%sa\\\n%s\n' "$line" "$stmt" >> "$script"
done
[ -s "$script" ] || { echo "trace_insert: no insertion point for $name in $points" >&2; exit 1; }
sed -f "$script" "$src"
