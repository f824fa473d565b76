#!/bin/bash
# Not collected by pytest: the drop-in script end to end on one GPU, every mode (synthetic data: the parquet directory is absent),
# plus a resume and the native sharded route.   bash tests/run_script_modes.sh   (from the repo root, on an MI355X box)
set -e -o pipefail
cd custom-yolo-implmentation_amd
out=${1:-/tmp/yolo_script_modes}
rm -rf "$out"; mkdir -p "$out"
sed "s#checkpoint_dir: .*#checkpoint_dir: \"$out/ck\"#" config.yaml > "$out/cfg.yaml"
python - "$out/cfg.yaml" "$out/cfg_native.yaml" <<'PY'
import sys, yaml
c = yaml.safe_load(open(sys.argv[1]))
c["training"]["fsdp2"]["native_shard"] = True
yaml.safe_dump(c, open(sys.argv[2], "w"))
PY
run() { echo "== $*"; python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) \
        scripts/distributed_training.py --device cuda --batch_size 8 "$@" 2>&1 | grep -E "Summary|Train -|Val   -|Saved checkpoint|Loaded model|Error|error" ; }
run --mode ddp --precision bfloat16 --config "$out/cfg.yaml"
run --mode ddp --precision float16 --config "$out/cfg.yaml"
run --mode fsdp2 --precision bfloat16 --config "$out/cfg.yaml"
run --mode fsdp --precision bfloat16 --config "$out/cfg.yaml"
run --mode fsdp2 --precision bfloat16 --config "$out/cfg_native.yaml"
last=$(ls -t "$out/ck" | head -1)
run --mode fsdp2 --precision bfloat16 --config "$out/cfg_native.yaml" --load_from_checkpoint "$last"
echo "all modes ran"
