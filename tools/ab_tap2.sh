bash tools/measure_round.sh r01f
