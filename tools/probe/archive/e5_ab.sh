# same-box A/B of the step over one GS_* switch: bash tools/probe/e5_ab.sh VAR a b [rounds]
var=$1; a=$2; b=$3; rounds=${4:-3}
for r in $(seq $rounds); do for v in $a $b; do echo -n "$var=$v "; env $var=$v python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done
