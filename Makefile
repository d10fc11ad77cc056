# Convenience targets; the driver uses __graft_entry__.build() / pytest / bench.py directly.
.PHONY: build test-cpu test-gpu bench window-loop sanitize clean
build:
	python -c "import __graft_entry__ as g; g.build()"
test-cpu: build
	python -m pytest tests -x -q -m "not gpu"
test-gpu: build            # needs an MI355X
	python -m pytest tests -x -q -m gpu
bench:                     # needs an MI355X
	python bench.py
window-loop:               # needs an MI355X: synthetic BAM -> .glf.txt -> VCF with the calls checked against the simulated variants
	python tools/n2_pipeline_bench.py --windows 60000 --vcf
sanitize: build            # ASan + UBSan over the CPU-side code, TSan over the window loop's threads; no GPU
	bash tests/sanitize_cpu.sh
clean:
	$(MAKE) -C dindel_tgi_amd/csrc clean
	$(MAKE) -C dindel_tgi_amd/host clean
	$(MAKE) -C oracle clean
