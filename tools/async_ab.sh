for h in 1 0; do for la in "" "--lookahead"; do
TSD_ICP_HELPERS=$h python bench.py --no-cpu-baseline --no-stream --async-mapping $la --no-second-pass 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('helpers $h la [$la] async value', round(d['value']), 'icp', round(d['ms_icp_iterate'],4), d['stages_ms'])"
done; done
