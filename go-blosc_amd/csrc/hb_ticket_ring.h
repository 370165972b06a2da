// hb_ticket_ring.h — results of hb_queue tickets whose slot was re-used before they were waited for (hipblosc.h: "its
// return value is kept for the newest 4 * depth such tickets").  Plain C++, no HIP: also built into the sanitizer check.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <utility>
#include <vector>

struct hb_ticket_ring {
    size_t cap = 0;
    std::vector<std::pair<int64_t, int64_t>> kept;      // {ticket, rc}, oldest first
    explicit hb_ticket_ring(size_t capacity = 0) : cap(capacity) {}
    void put(int64_t ticket, int64_t rc) {
        if (cap == 0) return;
        if (kept.size() >= cap) kept.erase(kept.begin());
        kept.push_back({ticket, rc});
    }
    // true + *rc when the ticket was kept (each ticket answers once)
    bool take(int64_t ticket, int64_t *rc) {
        for (size_t i = 0; i < kept.size(); i++)
            if (kept[i].first == ticket) { *rc = kept[i].second; kept.erase(kept.begin() + (long)i); return true; }
        return false;
    }
};
