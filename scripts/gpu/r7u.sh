#!/bin/bash
# round 6, builder side: the UNCHANGED reference entry script (eval_interactive_davis.py, byte for byte) through `python -m mivos_amd.dropin` on the final tree - the reference tree
# travels in the untracked, git-ignored scratch build/ref_tree (never committed; MIVOS_REFERENCE_ROOT), the driver's box has none and skips this test
set +e
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
rm -f gpurun_out/entry_script_parity.jsonl
MIVOS_REFERENCE_ROOT=$PWD/build/ref_tree timeout 900 python -m pytest tests/test_entry_script.py -q -m gpu -rs > gpurun_out/r7u_entry_script_pytest.txt 2>&1
echo "rc $?"; tail -6 gpurun_out/r7u_entry_script_pytest.txt | cut -c1-250
python -c "
import json
rows=[json.loads(l) for l in open('gpurun_out/entry_script_parity.jsonl')]
for r in rows: print({k: r[k] for k in list(r)[:8]})" 2>/dev/null | tail -4 | cut -c1-400
