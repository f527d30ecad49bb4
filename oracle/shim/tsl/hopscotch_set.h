#pragma once
#include <unordered_set>
