// Host side of the generic pat_str engine: the compiled program as vectors (tk_regex.cpp), uploaded once per Encoding by tk_create.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "tk_regex.h"

#define TK_RX_MAX_INS 448     // the kernels keep the program in LDS: 448 x 16 + 64 x 32 + 1024 x 8 + 64 x 32 bytes = 19 KiB
#define TK_RX_MAX_SETS 64
#define TK_RX_MAX_RANGES 1024  // pairs (a script is up to 176 of them, a binary property up to 632: tk_regex_binprops.inc)
#define TK_RX_MAX_FIRST 64    // first-byte bitmaps (32 bytes each)
#define TK_RX_DFA_MAX_STATES 4096u    // (ids are 15 bits)
#define TK_RX_DFA_MAX_ENTRIES 16384u  // states x classes: the kernels keep the transition table in LDS (32 KiB at most)

struct TkRxCompiled {
    std::vector<TkRxIns> ins;
    std::vector<TkRxSet> sets;
    std::vector<uint32_t> ranges;  // pairs (lo, hi)
    std::vector<uint32_t> first;   // first-byte bitmaps, 8 words each (TkRxProg::first)
    // the pattern as a DFA (tk_regex_dfa.inc; TkRxProg::dfa_*): empty when the pattern has none -- dfa_why says what stands in the way
    std::vector<uint16_t> dfa_trans;  // [dfa_nstates * dfa_ncls]
    std::vector<uint8_t> dfa_ascii;   // [128] classes of ASCII bytes, [128 + class] the class's group as look-behind / \b see it
    std::vector<uint16_t> dfa_s1;     // [0x1100]
    std::vector<uint8_t> dfa_s2;      // blocks of 256
    uint32_t dfa_ncls = 0, dfa_nstates = 0, dfa_flags = 0;  // flags bit 0: a match's start state depends on the char in front of it
    std::string dfa_why;
    bool has_dfa() const { return !dfa_trans.empty(); }
    bool empty() const { return ins.empty(); }
    // a view over the vectors and the built-in property table (host-side matching by the CPU tests)
    TkRxProg view() const;
};

// pat_str -> program; "" or the reason why the pattern is not supported
std::string tk_rx_compile(const char* pat_str, TkRxCompiled* out);
const uint8_t* tk_rx_props_stage1();  // [0x1100]
const uint8_t* tk_rx_props_stage2();  // [tk_rx_props_blocks() * 256]
uint32_t tk_rx_props_blocks();
