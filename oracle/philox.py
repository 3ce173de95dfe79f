"""Philox4x32-10 counter-based RNG (Salmon et al., SC'11 "Parallel random numbers:
as easy as 1, 2, 3"; Random123 reference constants) restated in pure Python.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference seeds nothing reproducibly on the hot path (SURVEY.md fact 7:
python ``random`` is never seeded; marlbase/dqn/model.py:105,109,113), so the HIP
path defines its own stream.  This file is the CPU statement of that stream; the
device code in codebase_amd/csrc/philox.h must produce identical words.

Stream layout used by the kernels (key = (seed_lo, seed_hi)):
  counter = (env_id, episode_idx, word_block, stream)
    stream 0 : per-step action noise; word_block = t (+ 2^16 * k for P > 3)
               word0 -> u (epsilon test), word1.. -> random action of agent 0..
    stream 1 : reset draws; sequential 32-bit words, word_block = draw_idx // 4
    stream 2 : replay sample indices (env_id := update idx, episode_idx := rank)
"""

M0 = 0xD2511F53
M1 = 0xCD9E8D57
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = 0xFFFFFFFF

STREAM_ACT = 0
STREAM_RESET = 1
STREAM_SAMPLE = 2


def philox4x32_10(ctr, key):
    """ctr: 4 uint32, key: 2 uint32 -> 4 uint32 (Random123 philox4x32-10)."""
    c0, c1, c2, c3 = (int(x) & MASK for x in ctr)
    k0, k1 = (int(x) & MASK for x in key)
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> 32, p0 & MASK
        hi1, lo1 = p1 >> 32, p1 & MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & MASK, lo1, (hi0 ^ c3 ^ k1) & MASK, lo0
        k0 = (k0 + W0) & MASK
        k1 = (k1 + W1) & MASK
    return c0, c1, c2, c3


def u01_f32(word):
    """float32 uniform in [0,1): top 24 bits * 2^-24 (exactly representable)."""
    return (word >> 8) * (1.0 / 16777216.0)


def bounded_nr(word, n):
    """Multiply-shift map of one 32-bit word onto [0, n) (no rejection).
    Used for the per-step random actions (bias <= n / 2^32)."""
    return (word * n) >> 32


class DrawStream:
    """Sequential 32-bit words of one (env, episode, stream) Philox stream, with
    an unbiased bounded-integer draw (Lemire 2019, multiply-shift + rejection)
    and the numpy-Generator-like surface the reset restatement consumes."""

    def __init__(self, seed, env_id, episode_idx, stream=STREAM_RESET):
        self.key = (seed & MASK, (seed >> 32) & MASK)
        self.env_id = env_id & MASK
        self.episode = episode_idx & MASK
        self.stream = stream
        self.idx = 0
        self._blk = None
        self._blk_no = -1

    def next_u32(self):
        b = self.idx >> 2
        if b != self._blk_no:
            self._blk = philox4x32_10((self.env_id, self.episode, b, self.stream), self.key)
            self._blk_no = b
        w = self._blk[self.idx & 3]
        self.idx += 1
        return w

    def integers(self, low, high):
        """Uniform integer in [low, high)."""
        n = int(high) - int(low)
        assert n > 0
        m = self.next_u32() * n
        lo = m & MASK
        if lo < n:
            t = ((1 << 32) - n) % n
            while lo < t:
                m = self.next_u32() * n
                lo = m & MASK
        return int(low) + (m >> 32)

    def permutation(self, n):
        """The device reset draws nothing for level permutations: all players /
        foods share one (min,max) level range on the C-ABI, so upstream's
        permutation of the bounds arrays cannot change the outcome."""
        return list(range(n))


def act_noise(seed, env, episode, t, n_agents, n_actions):
    """(u, [random action of agent p]) of one env-step - csrc/philox.h act_noise."""
    key = (seed & MASK, (seed >> 32) & MASK)
    words = []
    for k in range((1 + n_agents + 3) // 4):
        words += philox4x32_10((env, episode, t | (k << 16), STREAM_ACT), key)
    return u01_f32(words[0]), [bounded_nr(words[1 + p], n_actions) for p in range(n_agents)]
