// placeholder (NUTS device code follows)
#pragma once
