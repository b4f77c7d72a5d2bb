#!/bin/bash
# the data-dependence cases, two libraries side by side on one box: tools/dd_ab.sh lib_a lib_b
for ARGS in "" "--n-minors 1 --minor-frac 0.1 --minor-codons 45" "--n-minors 1 --minor-frac 0.3 --minor-codons 45" "--n-minors 1 --minor-frac 0.1 --minor-codons 300" "--n-minors 1 --minor-frac 0.3 --minor-codons 300" "--n-minors 2 --minor-frac 0.03 --minor-codons 300" "--n-minors 4 --minor-frac 0.2 --minor-codons 100" "--reads 125000 --force-dist --steps 200"; do
  echo "== $ARGS"; tools/ab.sh 1 "--verify $ARGS" "$@" || exit 1
done
