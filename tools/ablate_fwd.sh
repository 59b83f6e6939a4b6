#!/bin/bash
# timing-only ablations of the field forward (results are wrong by construction); run from the repo root here, then A/B on the GPU box
for v in "nomfma -DUMHS_ABL_NO_MFMA" "nostore -DUMHS_ABL_NO_STORE" "notrig -DUMHS_ABL_NO_TRIG" "nomfma_nostore -DUMHS_ABL_NO_MFMA -DUMHS_ABL_NO_STORE"; do
  set -- $v; name=$1; shift
  EXTRA="$*" bash tools/build_alt.sh $name >/dev/null && echo built $name
done
