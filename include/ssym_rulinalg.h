/*
 * ssym_rulinalg.h -- the ONE place that says how rulinalg 0.4.2's `utils::dot` combines its eight running sums.
 *
 * The reference's cosine_sim calls `rulinalg::utils::dot(&me[..len], &you[..len])` (src/sound.rs:31; crate pinned by
 * Cargo.toml:15, `rulinalg = "0.4.2"`).  The crate's source is NOT under /root/reference and could not be fetched in
 * this image, so the restatement below is from the published crate as remembered, and WHICH of the two associations
 * the crate uses could not be checked here:
 *
 *   loop over blocks of eight:  p_i = p_i + xs[i] * ys[i]             (i = 0..7; products and sums rounded separately)
 *   SSYM_RULINALG_COMBINE 0:    s = s + (p0 + p4);  s = s + (p1 + p5);  s = s + (p2 + p6);  s = s + (p3 + p7);
 *   SSYM_RULINALG_COMBINE 1:    s = s + p0 + p4;    ... i.e. s = (s + p0) + p4, left-associated
 *   tail (len mod 8):           s = s + xs[i] * ys[i]
 *
 * The two differ in the last ulp of the dot, i.e. in the bits of a similarity and -- only under near-ties -- in an
 * index.  Oracle (oracle/ssym_oracle.c, oracle/oracle.py) and product (csrc/refcos.hip: the tile kernel and the
 * one-query kernel; csrc/refcos_mfma.hip: the exact keys of the matrix-pipe search) all take the association from
 * this constant, so whoever holds the crate pins it with a one-line change (or -DSSYM_RULINALG_COMBINE=1 on both
 * builds); tools/rulinalg_variants.sh builds both and runs the refcos suite under each.
 */
#ifndef SSYM_RULINALG_H
#define SSYM_RULINALG_H

#ifndef SSYM_RULINALG_COMBINE
#define SSYM_RULINALG_COMBINE 0
#endif

/* one combine step with the caller's rounded addition ADD(x, y): s <- s (+) pa (+) pb in the chosen association */
#if SSYM_RULINALG_COMBINE == 0
#define SSYM_RULINALG_STEP(ADD, s, pa, pb) ADD((s), ADD((pa), (pb)))
#elif SSYM_RULINALG_COMBINE == 1
#define SSYM_RULINALG_STEP(ADD, s, pa, pb) ADD(ADD((s), (pa)), (pb))
#else
#error "SSYM_RULINALG_COMBINE must be 0 or 1"
#endif

#endif /* SSYM_RULINALG_H */
