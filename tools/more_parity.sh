mkdir -p gpurun_out/r4
python tools/decision_parity.py --songs 10000 --queries 2000 --snr 0 --workers 24 --config n640d64 --out gpurun_out/r4/decision_parity_cfg5_n640d64_f32_snr0.json 2>&1 | tail -1 | cut -c1-420
python tools/decision_parity.py --songs 10000 --queries 2000 --snr 0 --workers 24 --config seg --out gpurun_out/r4/decision_parity_segjson_snr0.json 2>&1 | tail -1 | cut -c1-420
for snr in -4 -2 0 2 4 6; do
python tools/decision_parity.py --songs 25000 --queries 2000 --snr $snr --workers 24 --out gpurun_out/r4/decision_parity_cfg3_snr$snr.json 2>&1 | tail -1 | cut -c1-420
done
