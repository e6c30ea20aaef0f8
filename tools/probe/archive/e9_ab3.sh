# same-box A/B/C of the step over one GS_* switch with three values
var=$1; shift
for r in 1 2 3; do for v in "$@"; do echo -n "$var=$v "; env $var=$v python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done
