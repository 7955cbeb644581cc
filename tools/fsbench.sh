set -x
mkdir -p gpurun_out/r06_fs
D="python tools/devbench.py --steps 400 --warmup 50"
for rep in 1 2; do
$D 
$D --stats none
$D --stats 3000:11000
$D --stats 0:16383
$D --stats 3000:11000 --rows-stats 1 --steps 100
$D --streams 3
$D --streams 3 --stats 3000:11000
$D --nfft 4096 --hop 4096 --frames 4096 --mode pow --hold 0
$D --nfft 4096 --hop 4096 --frames 4096 --mode pow --hold 0 --stats 500:3000
$D --nfft 8192 --hop 8192 --frames 8192 --hold 0 --steps 100
$D --nfft 8192 --hop 8192 --frames 8192 --hold 0 --steps 100 --stats 1000:7000
done
