import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import __graft_entry__ as ge
b = ge._load_binding()
import test_oracle_bvh as tob
print("calling", flush=True)
nodes, order, st = b.bvh_build_hlbvh(tob._hand_case_bounds(), 2)
print(st, flush=True)
tob.check_hand_case(nodes, order)
print("hand case ok", flush=True)
