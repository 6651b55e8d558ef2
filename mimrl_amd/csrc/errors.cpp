#include "common.h"
#include <cstdarg>
#include <cstdio>
namespace mimrl {
std::string& last_error_slot() { static thread_local std::string s; return s; }
int set_error(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  last_error_slot() = buf;
  return code;
}
}  // namespace mimrl
