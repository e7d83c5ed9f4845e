# interleaved A/B of two library variants on the paper's row lengths: tools/ab_small.sh <variantA> <variantB>
A=${1:-wave0}; B=${2:-wave1}
for cfg in "16384 1025 fwd 15 2.0" "4096 1025 lg 15 2.0" "65536 257 fwd 15 2.0" "16384 257 lg 15 2.0" "32768 513 fwd 15 2.0" "8192 2049 fwd 15 2.0" "65536 129 fwd 15 2.0" "16384 1025 fwd 8 1.0"; do
  set -- $cfg
  echo "== B=$1 N=$2 call=$3 flags=$4"
  AB_B=$1 AB_N=$2 AB_CALL=$3 AB_FLAGS=$4 AB_P=$5 AB_SETS=3 python tools/ab_probe.py $A $B 2>&1 | tail -2
done
