# Convenience targets; the driver uses __graft_entry__.build() / pytest / bench.py directly.
.PHONY: build test-cpu test-gpu bench clean
build:
	python -c "import __graft_entry__ as g; g.build()"
test-cpu: build
	python -m pytest tests -x -q -m "not gpu"
test-gpu: build            # needs an MI355X
	python -m pytest tests -x -q -m gpu
bench:                     # needs an MI355X
	python bench.py
clean:
	$(MAKE) -C dindel_tgi_amd/csrc clean
	$(MAKE) -C dindel_tgi_amd/host clean
	$(MAKE) -C oracle clean
