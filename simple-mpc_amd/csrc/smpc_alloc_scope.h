// smpc_alloc_scope.h -- device allocations of a constructor that throws are released.
//
// The engines allocate their buffers one after the other in their constructors; a failing allocation (or an argument check placed after
// the first of them) throws, no destructor runs, and everything allocated so far would leak.  An AllocScope at the top of a constructor
// records what dev_alloc hands out on this thread while it is alive and frees it again unless commit() was reached.
#pragma once
#include <algorithm>
#include <vector>

namespace smpc
{
  inline void dev_free(void * p); // (backend)
  struct AllocScope
  {
    static AllocScope *& current()
    {
      static thread_local AllocScope * cur = nullptr;
      return cur;
    }
    std::vector<void *> live;
    AllocScope * prev;
    bool committed = false;
    AllocScope() : prev(current()) { current() = this; }
    AllocScope(const AllocScope &) = delete;
    AllocScope & operator=(const AllocScope &) = delete;
    void commit() { committed = true; }
    ~AllocScope()
    {
      current() = prev;
      if (!committed)
      {
        std::vector<void *> v;
        v.swap(live);
        for (void * p : v)
          dev_free(p);
      }
    }
    static void note_alloc(void * p)
    {
      if (current() != nullptr && p != nullptr)
        current()->live.push_back(p);
    }
    static void note_free(void * p)
    {
      for (AllocScope * s = current(); s != nullptr; s = s->prev)
      {
        auto it = std::find(s->live.begin(), s->live.end(), p);
        if (it != s->live.end())
        {
          s->live.erase(it);
          return;
        }
      }
    }
  };
} // namespace smpc
