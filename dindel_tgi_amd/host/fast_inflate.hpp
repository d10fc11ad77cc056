// fast_inflate.hpp — a raw-DEFLATE (RFC 1951) decoder for BGZF blocks.
//
// The window loop's prepare stage spends more than half of its time in zlib's inflate (~0.15 ms of 0.27 ms per window on the
// synthetic samples of profiles/r02/n2_pipeline.md).  A BGZF block is small (<= 64 KiB out), complete in memory and followed by its
// CRC-32, which the reader checks anyway — so a decoder that keeps 64 bits of input in a register, looks symbols up in one table
// access for all common code lengths and copies matches eight bytes at a time can replace zlib there without any risk to the
// results: fastInflate() returns false for anything it does not like (then zlib decodes the block), and a wrong decode would fail
// the CRC like a corrupt file.  Written from the RFC; no zlib or libdeflate code.
#ifndef DINDEL_FAST_INFLATE_HPP
#define DINDEL_FAST_INFLATE_HPP
#include <cstddef>
#include <cstdint>

namespace dindel {

// Decodes the raw DEFLATE stream in[0, inLen) into out[0, outLen).  true iff the stream is well formed, ends with its final block
// inside the input and produces exactly outLen bytes.  `out` must have 8 bytes of slack behind outLen (writes may run that far over).
bool fastInflate(const uint8_t *in, size_t inLen, uint8_t *out, size_t outLen);

// CRC-32 (the gzip / BGZF polynomial) of buf[0, len), continuing from `crc` (0 to start): the value zlib's crc32() returns.  Uses
// carry-less multiplication (PCLMULQDQ folding, Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction",
// Intel 2009) when the CPU has it — 16 bytes per step instead of zlib 1.2.11's table walk, which runs no faster than its inflate on BAM
// data — and a byte-wise table otherwise.
uint32_t fastCrc32(uint32_t crc, const uint8_t *buf, size_t len);

} // namespace dindel
#endif
