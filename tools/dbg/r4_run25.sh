echo "budget 1 long 1: $(MOBGT_WGRAD_BUDGET=1 MOBGT_WGRAD_LONG=1 python tools/dbg/wgrad_group_bench.py 2>&1 | grep 'all six\|three long')"
echo "budget 0 slots 1024: $(MOBGT_WGRAD_BUDGET=0 MOBGT_WGRAD_SLOTS16=1024 python tools/dbg/wgrad_group_bench.py 2>&1 | grep 'all six\|three long')"
