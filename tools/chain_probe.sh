# settle-chain device times (ms) of the overlapped hand-manipulation reset against stream priorities / hardware queues:   sh tools/chain_probe.sh   (on the GPU box)
p() { echo "== $1"; python tools/host_profile_hand.py 2>&1 | grep "^chain\|^step kernel" | awk '{ if ($1=="chain") printf "%s ", $12; else print }' ; echo; }
GRX_CHAIN_STREAMS=6 p "6 chain streams"
GRX_CHAIN_STREAMS=2 p "2 chain streams"
GRX_CHAIN_STREAMS=4 p "4 chain streams"
GRX_CHAIN_STREAMS=5 p "5 chain streams"
