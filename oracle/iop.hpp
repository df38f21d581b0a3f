// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/field.hpp header).
//
// CPU restatement of the reference's IOP bookkeeping and BCS transformation, prover and verifier side, for BLAKE2b digests:
//   libiop/iop/iop.tcc                 registration state machine, rounds, virtual oracles, queries
//   libiop/bcs/bcs_common.tcc:399-696  Merkle trees per (round, domain), round parameters, hashchain per round
//   libiop/bcs/bcs_prover.tcc          signal_prover_round_done, proof of work, get_transcript
//   libiop/bcs/bcs_verifier.tcc        hashchain replay, set-membership validation, query responses from the transcript
// Independent of libiop_amd/: the device prover's transcript must equal this one's byte for byte.
// Citations are relative to /root/reference.
#pragma once
#include <functional>
#include <list>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <vector>
#include "domain.hpp"
#include "merkle.hpp"
#include "pow.hpp"

namespace oracle {

typedef std::vector<uint8_t> digest_t;

// ---- hashchain extractors (blake2b.tcc:162-257) ----
template<int W, uint64_t T>
std::vector<gf2n<W, T>> hashchain_squeeze(blake2b_hashchain &hc, size_t n, const gf2n<W, T> *)
{
    std::vector<gf2n<W, T>> out(n);
    if (n) hc.squeeze_binary_field(n, sizeof(gf2n<W, T>), (uint8_t *)out.data());
    else ++hc.squeeze_index;
    return out;
}
template<typename P>
std::vector<Fp<P>> hashchain_squeeze(blake2b_hashchain &hc, size_t n, const Fp<P> *)      // :187-227, :231-257
{
    ++hc.squeeze_index;
    uint8_t msg[DIGEST_LEN + 8];
    memcpy(msg, hc.state, DIGEST_LEN);
    memcpy(msg + DIGEST_LEN, &hc.squeeze_index, 8);
    std::vector<Fp<P>> out(n);
    for (size_t i = 0; i < n; ++i) {
        uint64_t key = i;
        while (true) {
            Fp<P> el;
            blake2b((uint8_t *)el.mont, sizeof(el.mont), msg, sizeof(msg), (const uint8_t *)&key, 8);
            size_t bitno = sizeof(el.mont) * 8 - 1;                                   // clear all bits above the modulus MSB
            while (!((P::modulus[bitno / 64] >> (bitno % 64)) & 1)) { el.mont[bitno / 64] &= ~((uint64_t)1 << (bitno % 64)); --bitno; }
            key += n;
            if (!Fp<P>::geq_mod(el.mont)) { out[i] = el; break; }
        }
    }
    return out;
}

// blake2b_field_element_hash (blake2b.tcc:140-160): BLAKE2b-256 of the raw object bytes
template<typename F> digest_t field_element_hash(const std::vector<F> &v)
{
    digest_t d(DIGEST_LEN);
    blake2b(d.data(), DIGEST_LEN, (const uint8_t *)v.data(), v.size() * sizeof(F));
    return d;
}

// ---- handles ----
struct oracle_handle { bool is_virtual; size_t id; };
struct position_handle { bool is_random; size_t id; };

template<typename F>
struct virtual_oracle {
    virtual ~virtual_oracle() {}
    virtual std::vector<F> evaluated_contents(const std::vector<const std::vector<F> *> &constituents) const = 0;
    virtual F evaluation_at_point(size_t position, const F &point, const std::vector<F> &constituents) const = 0;
};

// bcs_transformation_transcript (bcs_common.hpp:36-106)
template<typename F>
struct bcs_transcript {
    std::vector<std::vector<F>> prover_messages;
    std::vector<digest_t> MT_roots;
    std::vector<std::vector<size_t>> query_positions, MT_leaf_positions;
    std::vector<std::vector<std::vector<F>>> query_responses;          // [MT][position][oracle]
    std::vector<std::vector<digest_t>> MT_set_membership_proofs;       // auxiliary hashes (non-zk trees)
    digest_t proof_of_work;

    // canonical byte form used by the parity tests (the reference's own serialisation is text, bcs_common.tcc:96-390)
    std::vector<uint8_t> serialize() const
    {
        std::vector<uint8_t> out;
        auto u64 = [&](uint64_t v) { for (int i = 0; i < 8; ++i) out.push_back((uint8_t)(v >> (8 * i))); };
        auto raw = [&](const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; out.insert(out.end(), b, b + n); };
        u64(prover_messages.size());
        for (auto &m : prover_messages) { u64(m.size()); raw(m.data(), m.size() * sizeof(F)); }
        u64(MT_roots.size());
        for (auto &r : MT_roots) raw(r.data(), r.size());
        for (size_t t = 0; t < query_positions.size(); ++t) {      // one entry per tree; a holographic transcript has more trees than roots
            u64(query_positions[t].size());
            for (size_t p : query_positions[t]) u64(p);
            u64(MT_leaf_positions[t].size());
            for (size_t p : MT_leaf_positions[t]) u64(p);
            u64(query_responses[t].empty() ? 0 : query_responses[t][0].size());
            for (auto &col : query_responses[t]) raw(col.data(), col.size() * sizeof(F));
            u64(MT_set_membership_proofs[t].size());
            for (auto &h : MT_set_membership_proofs[t]) raw(h.data(), h.size());
        }
        raw(proof_of_work.data(), proof_of_work.size());
        return out;
    }
    // index_trees: the trees of the index round, whose roots a holographic transcript does not carry (bcs_prover.tcc:119-134)
    static bcs_transcript deserialize(const uint8_t *p, size_t len, size_t index_trees = 0)
    {
        size_t off = 0;
        auto need = [&](size_t n) { if (n > len - off) throw std::invalid_argument("truncated transcript"); };
        auto u64 = [&]() { need(8); uint64_t v; memcpy(&v, p + off, 8); off += 8; if (v > len) throw std::invalid_argument("implausible count"); return (size_t)v; };
        auto digest = [&]() { need(DIGEST_LEN); digest_t d(p + off, p + off + DIGEST_LEN); off += DIGEST_LEN; return d; };
        auto elems = [&](size_t n) { need(n * sizeof(F)); std::vector<F> v(n); if (n) memcpy((void *)v.data(), p + off, n * sizeof(F)); off += n * sizeof(F); return v; };
        bcs_transcript t;
        const size_t num_messages = u64();
        for (size_t i = 0; i < num_messages; ++i) { const size_t n = u64(); t.prover_messages.push_back(elems(n)); }
        const size_t num_roots = u64();
        for (size_t i = 0; i < num_roots; ++i) t.MT_roots.push_back(digest());
        for (size_t i = 0; i < num_roots + index_trees; ++i) {
            std::vector<size_t> qp(u64());
            for (size_t &v : qp) { need(8); uint64_t x; memcpy(&x, p + off, 8); off += 8; v = (size_t)x; }
            std::vector<size_t> lp(u64());
            for (size_t &v : lp) { need(8); uint64_t x; memcpy(&x, p + off, 8); off += 8; v = (size_t)x; }
            const size_t width = u64();
            std::vector<std::vector<F>> resp;
            for (size_t k = 0; k < qp.size(); ++k) resp.push_back(elems(width));
            std::vector<digest_t> aux(u64());
            for (digest_t &d : aux) d = digest();
            t.query_positions.push_back(qp); t.MT_leaf_positions.push_back(lp); t.query_responses.push_back(resp); t.MT_set_membership_proofs.push_back(aux);
        }
        t.proof_of_work = digest();
        if (off != len) throw std::invalid_argument("trailing bytes in transcript");
        return t;
    }
};

template<typename F>
class bcs_protocol {
public:
    typedef domain_of<F> D;
    typedef std::function<size_t(const std::vector<size_t> &)> position_calculator;

    // bcs_transformation_parameters (common_bcs_parameters.tcc:9-27, BLAKE2b): pow work parameter dim_h + 3, cost per hash 1
    explicit bcs_protocol(size_t pow_work_parameter) : pow_bitlen_(pow_bitlen(pow_work_parameter, 1)) {}
    bcs_protocol(size_t pow_work_parameter, const bcs_transcript<F> &transcript)
        : pow_bitlen_(pow_bitlen(pow_work_parameter, 1)), verifier_(true), transcript_(transcript) {}
    // preprocessing verifier (bcs_verifier.tcc:13-33): the index's roots (and messages) go in front of the transcript's
    bcs_protocol(size_t pow_work_parameter, const bcs_transcript<F> &transcript, const std::vector<digest_t> &index_MT_roots,
                 const std::vector<std::vector<F>> &indexed_messages = {})
        : pow_bitlen_(pow_bitlen(pow_work_parameter, 1)), verifier_(true), transcript_(transcript), num_index_roots_given_(index_MT_roots.size()),
          num_index_messages_given_(indexed_messages.size()), preprocessing_verifier_(true)
    {
        transcript_.MT_roots.insert(transcript_.MT_roots.begin(), index_MT_roots.begin(), index_MT_roots.end());
        transcript_.prover_messages.insert(transcript_.prover_messages.begin(), indexed_messages.begin(), indexed_messages.end());
    }

    // ---- registration (iop.tcc:22-263) ----
    size_t register_domain(const D &d) { domains_.push_back(d); return domains_.size() - 1; }
    const D &get_domain(size_t h) const { return domains_[h]; }

    oracle_handle register_oracle(size_t domain, size_t degree, bool make_zk)
    {
        assert_can_register(domain, degree);
        update_rounds_and_direction(true);
        if (is_holographic_ && num_interaction_rounds_ == 0) throw std::invalid_argument("Cannot register non-index oracles in round 0 of a holographic IOP");
        oracle_regs_.push_back({ domain, degree, make_zk });
        oracles_.emplace_back();
        oracles_present_.push_back(false);
        return { false, oracle_regs_.size() - 1 };
    }
    oracle_handle register_index_oracle(size_t domain, size_t degree)                                // iop.tcc:106-125
    {
        assert_can_register(domain, degree);
        if (num_prover_rounds_done_ != 0) throw std::invalid_argument("index oracles must be created in the 0th round");
        update_rounds_and_direction(true);
        is_holographic_ = true;
        oracle_regs_.push_back({ domain, degree, false });
        oracles_.emplace_back();
        oracles_present_.push_back(false);
        return { false, oracle_regs_.size() - 1 };
    }
    void signal_index_registrations_done()                                                           // iop.tcc:377-386
    {
        if (!is_holographic_ || num_interaction_rounds_ != 0) throw std::invalid_argument("Should only be used to end round 0 of a holographic IOP");
        update_rounds_and_direction(false);
    }
    bool is_holographic() const { return is_holographic_; }
    oracle_handle register_virtual_oracle(size_t domain, size_t degree, const std::vector<oracle_handle> &constituents,
                                          std::shared_ptr<virtual_oracle<F>> contents, bool cache_evaluated_contents = false)
    {
        assert_can_register(domain, degree);
        virtual_regs_.push_back({ domain, degree, constituents, contents, cache_evaluated_contents });
        virtual_point_cache_.emplace_back();
        return { true, virtual_regs_.size() - 1 };
    }
    size_t register_prover_message(size_t size)
    {
        update_rounds_and_direction(true);
        prover_message_sizes_.push_back(size);
        prover_messages_.emplace_back();
        prover_messages_present_.push_back(false);
        return prover_message_sizes_.size() - 1;
    }
    size_t register_verifier_random_message(size_t size)
    {
        update_rounds_and_direction(false);
        verifier_message_sizes_.push_back(size);
        return verifier_message_sizes_.size() - 1;
    }
    // bcs_common.tcc:482-495; round_parameters(domain): quotient_map_size_ = |domain| (bcs_common.hpp)
    void set_round_parameters(size_t quotient_map_size)
    {
        const size_t cur_round = num_interaction_rounds_;
        if (!round_params_.empty() && cur_round == round_params_.size() - 1) throw std::logic_error("Already set round parameters for this round");
        while (round_params_.size() < cur_round) round_params_.push_back(1);
        round_params_.push_back(quotient_map_size);
    }
    size_t get_round_parameters(size_t round) const { return round < round_params_.size() ? round_params_[round] : 1; }

    size_t get_oracle_degree(const oracle_handle &h) const { return h.is_virtual ? virtual_regs_[h.id].degree : oracle_regs_[h.id].degree; }

    void seal_interaction_registrations()                       // iop.tcc:227-251 + bcs_common.tcc:423-480 (+ bcs_verifier.tcc:36-105)
    {
        if (!from_prover_) throw std::logic_error("attempted to seal interaction registrations where verifier sends the last interactive message");
        num_oracles_at_end_of_round_.push_back(oracle_regs_.size());
        num_prover_messages_at_end_of_round_.push_back(prover_message_sizes_.size());
        ++num_interaction_rounds_;
        sealed_interactions_ = true;
        for (size_t round = 0; round < num_interaction_rounds_; ++round)
            for (auto &kv : oracles_in_round_by_domain(round)) {
                MT_num_leaves_.push_back(dom_size(domains_[kv.first]) / get_round_parameters(round));
                MT_nodes_.emplace_back();
            }
        if (verifier_) verifier_replay();
    }
    position_handle register_random_query_position(size_t domain) { random_position_domains_.push_back(domain); return { true, random_position_domains_.size() - 1 }; }
    position_handle register_deterministic_query_position(const std::vector<position_handle> &seeds, const position_calculator &fn)
    {
        deterministic_positions_.push_back({ seeds, fn });
        return { false, deterministic_positions_.size() - 1 };
    }
    size_t register_query(const oracle_handle &oracle, const position_handle &pos) { queries_.push_back({ oracle, pos }); return queries_.size() - 1; }
    void seal_query_registrations() {}

    size_t num_interaction_rounds() const { return num_interaction_rounds_; }

    // ---- prover (iop.tcc:265-433, bcs_prover.tcc:23-98) ----
    void submit_oracle(const oracle_handle &h, std::vector<F> &&contents)
    {
        if (oracles_present_[h.id]) throw std::invalid_argument("attempted to submit already submitted oracle");
        const size_t begin = num_prover_rounds_done_ == 0 ? 0 : num_oracles_at_end_of_round_[num_prover_rounds_done_ - 1];
        if (h.id < begin) throw std::invalid_argument("submitting an oracle for a previous round");
        if (h.id >= num_oracles_at_end_of_round_[num_prover_rounds_done_]) throw std::invalid_argument("submitting an oracle for a future round");
        if (dom_size(domains_[oracle_regs_[h.id].domain]) != contents.size()) throw std::invalid_argument("oracle evaluations don't match the domain size");
        oracles_[h.id] = std::move(contents);
        oracles_present_[h.id] = true;
    }
    void submit_prover_message(size_t h, std::vector<F> &&contents)
    {
        if (prover_messages_present_[h]) throw std::invalid_argument("attempted to submit already submitted prover message");
        if (prover_message_sizes_[h] != contents.size()) throw std::invalid_argument("prover message submission does not match its registered size");
        prover_messages_[h] = std::move(contents);
        prover_messages_present_[h] = true;
    }
    void signal_prover_round_done()
    {
        if (num_prover_rounds_done_ >= num_interaction_rounds_) throw std::logic_error("attempting to signal end of a round after protocol already finished");
        timed_block tb_round("Finish prover round");                                                 // bcs_prover.tcc:26
        const size_t ended_round = num_prover_rounds_done_;
        for (size_t id = min_oracle_id(ended_round); id < max_oracle_id(ended_round); ++id)
            if (!oracles_present_[id]) throw std::logic_error("signaling end of round without submitting all oracles in the round");
        const size_t mb = ended_round == 0 ? 0 : num_prover_messages_at_end_of_round_[ended_round - 1];
        for (size_t id = mb; id < num_prover_messages_at_end_of_round_[ended_round]; ++id)
            if (!prover_messages_present_[id]) throw std::logic_error("signaling end of round without submitting all prover messages in the round");
        ++num_prover_rounds_done_;
        // one tree per (round, domain) over all of that domain's oracles, leaves serialised by cosets (bcs_prover.tcc:36-47;
        // quirk F11: the reference indexes Merkle_trees_[processed_MTs_] for every domain of the round — one domain per round
        // in every shipped protocol, which is what is supported here)
        const auto mapping = oracles_in_round_by_domain(ended_round);
        if (mapping.size() > 1) throw std::logic_error("more than one oracle domain in a round (bcs_prover.tcc:36-47 would double-construct a tree)");
        const size_t cs = get_round_parameters(ended_round);
        std::vector<digest_t> roots;
        for (auto &kv : mapping) {
            std::vector<const uint8_t *> ptrs;
            for (size_t id : kv.second) ptrs.push_back((const uint8_t *)oracles_[id].data());
            const size_t n = dom_size(domains_[kv.first]);
            std::vector<uint8_t> &nodes = MT_nodes_[processed_MTs_];
            nodes.resize((2 * (n / cs) - 1) * DIGEST_LEN);
            timed_block tb_tree("Construct Merkle tree");                                             // bcs_prover.tcc:43
            merkle_build(ptrs.data(), ptrs.size(), sizeof(F), n, cs, dom_additive(domains_[kv.first]), nullptr, 0, nodes.data());
            roots.push_back(digest_t(nodes.begin(), nodes.begin() + DIGEST_LEN));
            ++processed_MTs_;
        }
        run_hashchain_for_round(ended_round, roots, prover_messages_);
        // bcs_prover.tcc:52-59; the indexer's one-round protocol has no proof of work (bcs_common.tcc:426-431, bcs_indexer.tcc:17-53)
        if (num_prover_rounds_done_ == num_interaction_rounds_ && !(is_holographic_ && num_interaction_rounds_ == 1)) {
            const digest_t challenge = squeeze_root_type();
            pow_answer_.resize(DIGEST_LEN);
            timed_block tb_pow("pow");                                                                // bcs_prover.tcc:52
            pow_solve_blake2b(challenge.data(), pow_bitlen_, pow_answer_.data());
        }
    }
    std::vector<F> obtain_verifier_random_message(size_t h) const
    {
        auto it = verifier_random_messages_.find(h);
        if (it == verifier_random_messages_.end()) throw std::logic_error("verifier random message not available yet");
        return it->second;
    }
    // iop.tcc:630-667
    const std::vector<F> &get_oracle_evaluations(const oracle_handle &h)
    {
        if (!h.is_virtual) return oracles_[h.id];
        auto cached = virtual_contents_cache_.find(h.id);
        if (cached != virtual_contents_cache_.end()) return cached->second;
        const virtual_reg &reg = virtual_regs_[h.id];
        std::vector<const std::vector<F> *> constituents;
        for (const oracle_handle &c : reg.constituents) constituents.push_back(&get_oracle_evaluations(c));
        std::vector<F> result = reg.contents->evaluated_contents(constituents);
        if (reg.cache) return virtual_contents_cache_[h.id] = std::move(result);
        scratch_.push_back(std::move(result));                                             // non-cached: kept alive for the caller
        return scratch_.back();
    }
    void drop_scratch() { scratch_.clear(); }

    // bcs_prover.tcc:136-233
    bcs_transcript<F> get_transcript()
    {
        bcs_transcript<F> result;
        result.prover_messages = prover_messages_;
        for (auto &nodes : MT_nodes_) result.MT_roots.push_back(digest_t(nodes.begin(), nodes.begin() + DIGEST_LEN));
        for (size_t q = 0; q < queries_.size(); ++q) obtain_query_response(q);
        size_t MT_idx = 0;
        for (size_t round = 0; round < num_interaction_rounds_; ++round) {
            const size_t cs = get_round_parameters(round);
            for (auto &kv : oracles_in_round_by_domain(round)) {
                std::set<size_t> qset, lset;
                const size_t n = dom_size(domains_[kv.first]), num_leaves = n / cs;
                for (size_t id : kv.second)
                    for (size_t pos : oracle_id_to_query_positions_[id]) {
                        qset.insert(pos);
                        lset.insert(query_position_to_merkle_tree_position(pos, num_leaves, cs, dom_additive(domains_[kv.first])));
                    }
                std::vector<size_t> qpos(qset.begin(), qset.end()), lpos(lset.begin(), lset.end());
                std::vector<std::vector<F>> values;
                for (size_t pos : qpos) {
                    std::vector<F> column;
                    for (size_t id : kv.second) column.push_back(oracles_[id][pos]);
                    values.push_back(column);
                }
                result.query_positions.push_back(qpos);
                result.MT_leaf_positions.push_back(lpos);
                result.query_responses.push_back(values);
                std::vector<digest_t> aux;
                for (size_t node : membership_proof_node_indices(num_leaves, lpos))
                    aux.push_back(digest_t(MT_nodes_[MT_idx].begin() + node * DIGEST_LEN, MT_nodes_[MT_idx].begin() + (node + 1) * DIGEST_LEN));
                result.MT_set_membership_proofs.push_back(aux);
                ++MT_idx;
            }
        }
        if (is_holographic_) {                                                             // remove_index_info_from_transcript (bcs_prover.tcc:119-134)
            const size_t index_trees = oracles_in_round_by_domain(0).size();
            result.prover_messages.erase(result.prover_messages.begin(), result.prover_messages.begin() + num_prover_messages_at_end_of_round_[0]);
            result.MT_roots.erase(result.MT_roots.begin(), result.MT_roots.begin() + index_trees);
        }
        result.proof_of_work = pow_answer_;
        return result;
    }
    // bcs_indexer.tcc:67-77: what the verifier keeps of the index
    std::vector<digest_t> get_index_MT_roots() const
    {
        std::vector<digest_t> roots;
        for (size_t i = 0; i < processed_MTs_ && i < oracles_in_round_by_domain(0).size(); ++i) roots.push_back(digest_t(MT_nodes_[i].begin(), MT_nodes_[i].begin() + DIGEST_LEN));
        return roots;
    }
    std::vector<F> take_oracle(const oracle_handle &h) { return std::move(oracles_[h.id]); }

    // ---- shared by prover and verifier (iop.tcc:480-565, 669-714) ----
    size_t obtain_query_position(const position_handle &p)
    {
        if (p.is_random) {
            auto it = random_positions_.find(p.id);
            if (it != random_positions_.end()) return it->second;
            const size_t n = dom_size(domains_[random_position_domains_[p.id]]);
            return random_positions_[p.id] = hashchain_.squeeze_query_positions(1, n)[0];     // bcs_common.tcc:536-548
        }
        auto it = deterministic_position_values_.find(p.id);
        if (it != deterministic_position_values_.end()) return it->second;
        std::vector<size_t> seeds;
        for (const position_handle &s : deterministic_positions_[p.id].seeds) seeds.push_back(obtain_query_position(s));
        return deterministic_position_values_[p.id] = deterministic_positions_[p.id].fn(seeds);
    }
    F obtain_query_response(size_t q)
    {
        auto it = query_responses_.find(q);
        if (it != query_responses_.end()) return it->second;
        const size_t pos = obtain_query_position(queries_[q].position);
        return query_responses_[q] = get_oracle_evaluation_at_point(queries_[q].oracle, pos);
    }
    F get_oracle_evaluation_at_point(const oracle_handle &h, size_t position)
    {
        if (!h.is_virtual) {
            if (position >= dom_size(domains_[oracle_regs_[h.id].domain])) throw std::invalid_argument("evaluation position is outside of domain");
            if (verifier_) {                                                              // bcs_verifier.tcc:190-215
                auto it = oracle_id_and_pos_to_value_.find({ h.id, position });
                if (it == oracle_id_and_pos_to_value_.end()) throw std::logic_error("query position not present in the transcript");
                return it->second;
            }
            oracle_id_to_query_positions_[h.id].insert(position);
            return oracles_[h.id][position];
        }
        auto &cache = virtual_point_cache_[h.id];
        auto it = cache.find(position);
        if (it != cache.end()) return it->second;
        const virtual_reg &reg = virtual_regs_[h.id];
        std::vector<F> constituents;
        for (const oracle_handle &c : reg.constituents) constituents.push_back(get_oracle_evaluation_at_point(c, position));
        const F point = dom_element(domains_[reg.domain], position);
        return cache[position] = reg.contents->evaluation_at_point(position, point, constituents);
    }

    // ---- verifier ----
    bool transcript_is_valid() const { return transcript_is_valid_; }
    std::vector<F> receive_prover_message(size_t h) const { return verifier_ ? transcript_.prover_messages[h] : prover_messages_[h]; }

    std::map<size_t, std::vector<size_t>> oracles_in_round_by_domain(size_t round) const   // iop.tcc:801-820
    {
        std::map<size_t, std::vector<size_t>> mapping;
        for (size_t id = min_oracle_id(round); id < max_oracle_id(round); ++id) mapping[oracle_regs_[id].domain].push_back(id);
        return mapping;
    }

private:
    struct oracle_reg { size_t domain, degree; bool make_zk; };
    struct virtual_reg { size_t domain, degree; std::vector<oracle_handle> constituents; std::shared_ptr<virtual_oracle<F>> contents; bool cache; };
    struct det_position { std::vector<position_handle> seeds; position_calculator fn; };
    struct query_reg { oracle_handle oracle; position_handle position; };

    void assert_can_register(size_t domain, size_t degree) const                           // iop.tcc:65-84
    {
        if (sealed_interactions_) throw std::logic_error("attempted to register an oracle after interactive registrations sealed");
        if (domain >= domains_.size()) throw std::invalid_argument("domain not registered");
        if (degree >= dom_size(domains_[domain])) throw std::invalid_argument("attempting to register oracle whose degree exceeds domain size");
    }
    void update_rounds_and_direction(bool new_from_prover)                                 // iop.tcc:36-63
    {
        if (sealed_interactions_) throw std::logic_error("registration after interactive registrations sealed");
        if (from_prover_ == new_from_prover) return;
        if (from_prover_) {
            num_oracles_at_end_of_round_.push_back(oracle_regs_.size());
            num_prover_messages_at_end_of_round_.push_back(prover_message_sizes_.size());
            num_interaction_rounds_ += 1;
        } else {
            num_verifier_messages_at_end_of_round_.push_back(verifier_message_sizes_.size());
        }
        from_prover_ = new_from_prover;
    }
    size_t min_oracle_id(size_t round) const { return round == 0 ? 0 : num_oracles_at_end_of_round_[round - 1]; }
    size_t max_oracle_id(size_t round) const { return num_oracles_at_end_of_round_[round]; }

    static size_t query_position_to_merkle_tree_position(size_t pos, size_t num_leaves, size_t cs, bool additive)   // bcs_common.tcc:682-696
    {
        if (cs == 1) return pos;
        return additive ? pos / cs : pos % num_leaves;
    }

    // bcs_common.tcc:550-614
    void run_hashchain_for_round(size_t round, const std::vector<digest_t> &roots, const std::vector<std::vector<F>> &all_prover_messages)
    {
        for (const digest_t &r : roots) hashchain_.absorb_digest(r.data());
        const size_t min_id = round == 0 ? 0 : num_prover_messages_at_end_of_round_[round - 1];
        const size_t max_id = num_prover_messages_at_end_of_round_[round];
        std::vector<F> concat = { F::zero() };
        for (size_t id = min_id; id < max_id; ++id) concat.insert(concat.end(), all_prover_messages[id].begin(), all_prover_messages[id].end());
        hashchain_.absorb_digest(field_element_hash<F>(concat).data());                     // blake2b.tcc:68-74 (the input is ignored, F8)
        const size_t start = num_verifier_messages_at_end_of_round_[round];
        const size_t end = (round == num_interaction_rounds_ - 1) ? 0 : num_verifier_messages_at_end_of_round_[round + 1];
        for (size_t i = start; i < end; ++i)
            verifier_random_messages_[i] = hashchain_squeeze(hashchain_, verifier_message_sizes_[i], (const F *)nullptr);
    }
    digest_t squeeze_root_type()                                                           // blake2b.tcc:105-110
    {
        return field_element_hash<F>(hashchain_squeeze(hashchain_, 1, (const F *)nullptr));
    }

    // merkle_tree::serialize_leaf_values_by_coset (merkle_tree.tcc:153-198)
    std::vector<std::vector<F>> serialize_leaf_values_by_coset(const std::vector<size_t> &qpos, const std::vector<std::vector<F>> &responses,
                                                               size_t cs, size_t num_leaves, bool additive) const
    {
        std::vector<std::vector<F>> columns(qpos.size() / cs);
        const size_t leaf_size = responses.empty() ? 0 : responses[0].size() * cs;
        for (auto &c : columns) c.assign(leaf_size, F::zero());
        std::vector<size_t> intra(columns.size(), 0);
        std::map<size_t, size_t> leaf_to_index;
        size_t next_index = 0;
        for (size_t i = 0; i < qpos.size(); ++i) {
            const size_t leaf = coset_index(additive, num_leaves * cs, qpos[i], cs);
            if (!leaf_to_index.count(leaf)) leaf_to_index[leaf] = next_index++;
            const size_t r = leaf_to_index[leaf];
            if (r >= columns.size() || intra[r] >= cs) throw std::invalid_argument("query positions do not form whole cosets");
            const size_t in_coset = intra[r]++;
            for (size_t j = 0; j < responses[i].size(); ++j) columns[r][j * cs + in_coset] = responses[i][j];
        }
        return columns;
    }

    // bcs_verifier.tcc:36-140
    void verifier_replay()
    {
        transcript_is_valid_ = true;
        size_t processed = 0;
        try {
            const size_t num_MTs = MT_num_leaves_.size();
            if (transcript_.MT_roots.size() != num_MTs || transcript_.query_positions.size() != num_MTs || transcript_.MT_leaf_positions.size() != num_MTs ||
                transcript_.query_responses.size() != num_MTs || transcript_.MT_set_membership_proofs.size() != num_MTs)
                throw std::invalid_argument("transcript does not hold one entry per Merkle tree");
            for (size_t t = 0; t < num_MTs; ++t)
                if (transcript_.query_responses[t].size() != transcript_.query_positions[t].size()) throw std::invalid_argument("one response column per query position");
            for (size_t round = 0; round < num_interaction_rounds_; ++round) {
                const auto mapping = oracles_in_round_by_domain(round);
                const size_t num_domains = mapping.size();
                if (preprocessing_verifier_ && round == 0) {                                 // bcs_verifier.tcc:47-59
                    if (num_domains != num_index_roots_given_) throw std::invalid_argument("Index had an incorrect number of MT roots");
                    if (num_prover_messages_at_end_of_round_[0] != num_index_messages_given_) throw std::invalid_argument("Index had an incorrect number of prover messages");
                }
                if (processed + num_domains > transcript_.MT_roots.size()) throw std::invalid_argument("transcript has too few Merkle roots");
                std::vector<digest_t> roots(transcript_.MT_roots.begin() + processed, transcript_.MT_roots.begin() + processed + num_domains);
                if (transcript_.prover_messages.size() != prover_message_sizes_.size()) throw std::invalid_argument("transcript has the wrong number of prover messages");
                for (size_t i = 0; i < prover_message_sizes_.size(); ++i)
                    if (transcript_.prover_messages[i].size() != prover_message_sizes_[i]) throw std::invalid_argument("prover message of the wrong size");
                run_hashchain_for_round(round, roots, transcript_.prover_messages);
                const size_t cs = get_round_parameters(round);
                for (auto &kv : mapping) {
                    const bool additive = dom_additive(domains_[kv.first]);
                    const size_t num_leaves = MT_num_leaves_[processed];
                    const auto &qpos = transcript_.query_positions[processed];
                    const auto &lpos = transcript_.MT_leaf_positions[processed];
                    const auto &resp = transcript_.query_responses[processed];
                    for (auto &col : resp) if (col.size() != kv.second.size()) throw std::invalid_argument("query response of the wrong width");
                    const std::vector<std::vector<F>> columns = cs == 1 ? resp : serialize_leaf_values_by_coset(qpos, resp, cs, num_leaves, additive);
                    if (columns.size() != lpos.size()) throw std::invalid_argument("leaf positions do not match the query responses");
                    std::vector<digest_t> leaf_hashes;
                    for (auto &c : columns) leaf_hashes.push_back(field_element_hash<F>(c));
                    for (size_t i = 0; i + 1 < lpos.size(); ++i) if (lpos[i] >= lpos[i + 1]) throw std::invalid_argument("leaf positions must be sorted and unique");
                    for (size_t p : lpos) if (p >= num_leaves) throw std::invalid_argument("leaf position out of range");
                    bool ok = !lpos.empty();
                    if (ok) ok = membership_proof_validate(roots_at(processed), num_leaves, lpos, leaf_hashes, transcript_.MT_set_membership_proofs[processed]);
                    if (!ok) transcript_is_valid_ = false;
                    // parse_query_responses_from_transcript (bcs_verifier.tcc:108-140)
                    size_t k = 0;
                    for (size_t id : kv.second) {
                        for (size_t i = 0; i < qpos.size(); ++i) oracle_id_and_pos_to_value_[{ id, qpos[i] }] = resp[i][k];
                        ++k;
                    }
                    ++processed;
                }
            }
            const digest_t challenge = squeeze_root_type();
            if (transcript_.proof_of_work.size() != DIGEST_LEN || !pow_verify_blake2b(challenge.data(), transcript_.proof_of_work.data(), pow_bitlen_))
                transcript_is_valid_ = false;
        } catch (const std::exception &) {
            transcript_is_valid_ = false;
        }
    }
    const uint8_t *roots_at(size_t i) const { return transcript_.MT_roots[i].data(); }

    // registration state
    std::vector<D> domains_;
    std::vector<oracle_reg> oracle_regs_;
    std::vector<virtual_reg> virtual_regs_;
    std::vector<size_t> prover_message_sizes_, verifier_message_sizes_;
    std::vector<size_t> num_oracles_at_end_of_round_, num_prover_messages_at_end_of_round_, num_verifier_messages_at_end_of_round_;
    std::vector<size_t> round_params_;
    bool from_prover_ = false, sealed_interactions_ = false, is_holographic_ = false;
    size_t num_interaction_rounds_ = 0;
    std::vector<size_t> random_position_domains_;
    std::vector<det_position> deterministic_positions_;
    std::vector<query_reg> queries_;
    // run state
    std::vector<std::vector<F>> oracles_, prover_messages_;
    std::vector<bool> oracles_present_, prover_messages_present_;
    size_t num_prover_rounds_done_ = 0, processed_MTs_ = 0;
    std::vector<size_t> MT_num_leaves_;
    std::vector<std::vector<uint8_t>> MT_nodes_;
    blake2b_hashchain hashchain_;
    std::map<size_t, std::vector<F>> verifier_random_messages_;
    std::map<size_t, std::vector<F>> virtual_contents_cache_;
    std::list<std::vector<F>> scratch_;        // non-cached virtual-oracle contents handed out by reference
    std::vector<std::map<size_t, F>> virtual_point_cache_;
    std::map<size_t, size_t> random_positions_, deterministic_position_values_;
    std::map<size_t, F> query_responses_;
    std::map<size_t, std::set<size_t>> oracle_id_to_query_positions_;
    size_t pow_bitlen_;
    digest_t pow_answer_;
    // verifier state
    bool verifier_ = false, transcript_is_valid_ = false;
    bcs_transcript<F> transcript_;
    size_t num_index_roots_given_ = 0, num_index_messages_given_ = 0;
    bool preprocessing_verifier_ = false;
    std::map<std::pair<size_t, size_t>, F> oracle_id_and_pos_to_value_;
};

} // namespace oracle
