cd "$GRAFT_REPO_ROOT"
run() {
  python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-variants --no-check "$@" 2>/dev/null | python -c '
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=r["stages"]
print("  ms/step %.4f device %.4f | "%(r["ms_per_step"],r["device_ms_per_step"])+" ".join("%s %.4f (%.2f GB)"%(k.replace("coarse_",""),v["ms_per_step"],v["necessary_gb_per_step"] or 0) for k,v in s.items()))'
}
echo "== product"; run; run
echo "== --no-carry"; run --no-carry; run --no-carry
tools/build_variant.sh plain -DGA_PREMIX_PLAIN=1 >/dev/null 2>&1
echo "== plain sum"; run --library tools/variants/plain.so; run --library tools/variants/plain.so
echo "== plain sum, no carry"; run --library tools/variants/plain.so --no-carry; run --library tools/variants/plain.so --no-carry
