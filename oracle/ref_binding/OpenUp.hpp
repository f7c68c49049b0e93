// oracle/ref_binding/OpenUp.hpp -- TEST INFRASTRUCTURE.  Included by the binding's translation units BEFORE they open the
// reference's class definitions up (#define private public / protected public / class struct): every standard and third-party
// header the reference's headers pull in, so that the macros only ever touch the reference's own declarations.
#ifndef OPENUP_HPP_
#define OPENUP_HPP_
#include <algorithm>
#include <array>
#include <atomic>
#include <cassert>
#include <cctype>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <csetjmp>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <exception>
#include <fstream>
#include <functional>
#include <iomanip>
#include <iosfwd>
#include <iostream>
#include <iterator>
#include <limits>
#include <locale>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <ostream>
#include <regex>
#include <set>
#include <sstream>
#include <stack>
#include <stdexcept>
#include <streambuf>
#include <string>
#include <thread>
#include <type_traits>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>
#include <emmintrin.h>
#include <immintrin.h>
#include <embree2/rtcore.h>
#include <embree2/rtcore_ray.h>
#include <embree2/rtcore_scene.h>
#include <embree2/rtcore_geometry.h>
#include <rapidjson/document.h>
#include <rapidjson/prettywriter.h>
#include <rapidjson/stringbuffer.h>
#include <rapidjson/writer.h>
#include <tinyformat/tinyformat.hpp>
#include <sobol/sobol.h>
// the two reference headers that spell a template parameter `class T` (the macro would turn it into `struct T`); their include
// guards make the later inclusions no-ops
#include "Memory.hpp"
#include "AlignedAllocator.hpp"
#endif
