#!/bin/bash
# The host decoders' inner loop on the real query stream of an 8 MiB text archive, on THIS machine's CPU (no GPU work):
# builds tools/decoder_replay.cpp with the product's compiler, makes the archive with the oracle, replays it; then
# the marginal cost of the query classes (REPLACE_*: those queries turned into binary ones).
cd "$(dirname "$0")/.."
mkdir -p tools/_build
/opt/rocm/lib/llvm/bin/clang++ -O3 -std=c++17 -I bce_amd/csrc tools/decoder_replay.cpp bce_amd/csrc/host_coder.cpp -o tools/_build/decreplay -lpthread || exit 1
python3 - <<'PY'
import oracle
open('/tmp/bce_replay_t8.bce', 'wb').write(oracle.compress(oracle.synth_text(1, 8 << 20)))
PY
lscpu | grep "Model name"
echo "== the decoders as built"; tools/_build/decreplay /tmp/bce_replay_t8.bce 7 | tail -3
for v in REPLACE_WIDE REPLACE_ESC REPLACE_MID; do echo -n "[$v] "; env $v=1 tools/_build/decreplay /tmp/bce_replay_t8.bce 5 | tail -1; done
