#!/bin/bash
# Wide-MLP (8 x 8192, batch 8192, bf16) on one GPU: plain, under the reducer with one-rank RCCL, and under the
# reducer with the emulated 8-rank all-reduce load at several bus bandwidths / footprints / bucket sizes.
out=gpurun_out/wm3
mkdir -p $out
b() {  # name, args...
  name=$1; shift
  timeout -k 10 170 python bench.py --model widemlp --steps 30 --warmup 8 "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -5 $out/$name.err; exit 1; }
  python - "$out/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
keys = ("ms_per_step", "comm_busy_ms", "exposed_comm_ms", "collectives_per_step", "grads_written_in_place", "grads_packed_by_copy")
print(sys.argv[2], {k: d.get(k) for k in keys}, d["config"].get("comm_stream"), flush=True)
PY
}
case "${1:-all}" in
base)
  b plain
  b ddp1 --force-ddp
  b ddp1_high --force-ddp
  ;;
emu)
  for bw in 150 300 500; do
    b emu8_bw${bw}_wg32 --force-ddp --emulate-ranks 8 --emulate-busbw $bw --emulate-wgs 32
  done
  for wg in 16 64; do
    b emu8_bw300_wg${wg} --force-ddp --emulate-ranks 8 --emulate-busbw 300 --emulate-wgs $wg
  done
  for mb in 32 128 256; do
    b emu8_bw300_wg32_bucket${mb} --force-ddp --emulate-ranks 8 --emulate-busbw 300 --emulate-wgs 32 --bucket-mb $mb --max-collective-mb $mb
  done
  ;;
esac
if [ "${1:-all}" = "gpt2" ]; then
  g() { name=$1; shift
    timeout -k 10 280 python bench.py --steps 8 --warmup 3 "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -5 $out/$name.err; exit 1; }
    python - "$out/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
keys = ("value", "ms_per_step", "comm_busy_ms", "exposed_comm_ms", "collectives_per_step", "grads_written_in_place", "grads_packed_by_copy", "peak_mem_gb")
print(sys.argv[2], {k: d.get(k) for k in keys}, flush=True)
PY
  }
  g gpt2_plain
  g gpt2_emu8_bw300 --force-ddp --emulate-ranks 8 --emulate-busbw 300
  g gpt2_emu8_bw150 --force-ddp --emulate-ranks 8 --emulate-busbw 150
fi
