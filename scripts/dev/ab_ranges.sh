for cfg in "rb1024 256" "rb512 512" "rb512 768" "rb256 1024" "rb256 1536"; do set -- $cfg; export SGDNET_BIN_RANGES=$2; echo "ranges $2:"; scripts/dev/ab_variants.sh "--workload C5s" $1; done
