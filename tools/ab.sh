#!/bin/bash
# usage: tools/ab.sh "A B C" "0 3"   -- swap in tools/libvcmi_<name>.so and run the variant sweep
for n in $1; do
  cp tools/libvcmi_$n.so voiceconversion.jl_amd/libvcmi.so
  echo "== lib $n"
  ./tools/variants.sh "$2"
done
